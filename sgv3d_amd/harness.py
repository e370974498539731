"""The reference Lightning module's ``eval_step`` as a plain function, for measuring and testing the drop-in.

exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:242-258 is what a user who only swaps the ``BEVHeight``
import runs per validation batch: every calibration tensor of the batch goes through ``.cuda()`` (a FRESH device tensor
per step), one eager ``self.model(sweep_imgs, mats)``, ``get_bboxes``, and three ``.detach().cpu().numpy()`` per sample
(each a host synchronisation).  Nothing here knows about ``FramePipeline`` or any other API of this build: the speed of
this function is the speed the unchanged harness gets (``bench.py``: ``harness_eval_step``).
"""
import torch


def eval_step(model, batch):
    """``BEVHeightLightningModel.eval_step`` (exps/...:242-258) with ``self.model`` = ``model``; ``batch`` is the tuple
    the reference's collate function yields: ``(sweep_imgs, mats, _, img_metas, _, _)``.  The DistributedDataParallel
    branch of the original is dead under Lightning and is not restated."""
    imgs, host_mats, _, metas, _, _ = batch
    mats = {k: v.cuda() for k, v in host_mats.items()}        # :245-246 -- a fresh device tensor per entry, every step
    imgs = imgs.cuda()                                        # :247
    preds = model(imgs, mats)                                 # :248
    out = []
    for (boxes, scores, labels), meta in zip(model.get_bboxes(preds, metas), metas):       # :252, :253-257
        out.append([boxes.tensor.detach().cpu().numpy(), scores.detach().cpu().numpy(), labels.detach().cpu().numpy(), meta])
    return out


def make_batch(imgs, host_mats, img_metas=None):
    """A batch as the data loader hands it over: calibration tensors on the HOST (new dict per step, as the loader's
    collate makes one), images wherever the caller keeps them (``bench.py`` keeps them resident in HBM, for which
    ``.cuda()`` is the identity; a CPU tensor pays the H2D copy inside ``eval_step`` like in the reference)."""
    B = int(imgs.shape[0])
    metas = img_metas if img_metas is not None else [{'token': f'frame{i}'} for i in range(B)]
    return (imgs, dict(host_mats), None, metas, None, None)
