"""ORACLE — TEST INFRASTRUCTURE ONLY.  numpy restatement of the lift step
(reference: layers/backbones/lss_fpn.py:462-466 softmax (x) context, :469-486 reshape/permute)."""
import numpy as np


def lift(height_feature, D, C):
    """height_feature f32 [B, D+C, fH, fW] (NCHW, height logits first)
    -> prob f32 [B, D, fH, fW], lifted f32 [B, 1, D, fH, fW, C]."""
    hf = np.asarray(height_feature, np.float32)
    logits = hf[:, :D]
    m = logits.max(axis=1, keepdims=True)
    e = np.exp(logits - m, dtype=np.float32)
    prob = (e / e.sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)   # softmax(1), :462
    ctx = hf[:, D:D + C]
    prod = prob[:, None] * ctx[:, :, None]                       # [B, C, D, fH, fW], :464-466
    lifted = np.ascontiguousarray(prod.transpose(0, 2, 3, 4, 1))[:, None]   # permute(0,1,3,4,5,2), :486
    return prob, lifted
