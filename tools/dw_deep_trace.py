#!/usr/bin/env python3
"""GPU durations (rocprofv3 kernel trace, not host timing) of the plain and the two-chunks-ahead 64x256 tile of the direct-weight
kernel on cfg-5 batch-1 layers.  Run under: rocprofv3 --kernel-trace --output-format csv -d DIR -o dw -- python3 tools/dw_deep_trace.py;
then: python3 tools/dw_deep_trace.py DIR/dw_kernel_trace.csv"""
import os, sys
SHAPES = [(1024, 54, 96, 256, 1, 1, 0, 1), (256, 54, 96, 1024, 1, 1, 0, 1), (256, 54, 96, 256, 3, 1, 1, 1), (512, 108, 192, 128, 1, 1, 0, 1),
          (2048, 27, 48, 512, 1, 1, 0, 1), (512, 27, 48, 512, 3, 1, 1, 1), (512, 54, 96, 512, 3, 1, 1, 1), (2560, 54, 96, 512, 1, 1, 0, 1)]
REPS = 10
if len(sys.argv) > 1:
    import csv
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv_dw_bf16_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
    i = 0
    for sh in SHAPES:
        line = f"{sh[0]}->{sh[3]} k{sh[4]} @{sh[1]}x{sh[2]}: "
        for t in (31, 36, 38, 39):
            seg = sorted(d[i + 1:i + 1 + REPS]); i += 1 + REPS          # (first launch = warm-up)
            line += f"tile {t} {seg[len(seg) // 2]:6.1f} us   "
        print(line)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgv3d_amd import hip_ops
hip_ops.MFMA_BF16 = True
for cin, H, W, cout, k, stride, pad, dil in SHAPES:
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    conv = hip_ops.PackedConv(w, stride=stride, pad=pad, dil=dil, relu=True)
    x = torch.randn(1, H, W, cin, device="cuda").bfloat16()
    out = torch.empty(1, *conv.out_hw(H, W), cout, dtype=torch.bfloat16, device="cuda")
    for t in (31, 36, 38, 39):
        for _ in range(1 + REPS):
            conv(x, out, tile=t, split_k=1)
        torch.cuda.synchronize()
