#!/usr/bin/env python3
"""HBM traffic per LAYER of one cfg-2 forward: joins the per-dispatch FETCH_SIZE / WRITE_SIZE of two rocprofv3 --pmc runs of
tools/layer_report.py (dispatch order of the last frame) with the layer list that run wrote (gpurun_out/layers.json) and
prints measured bytes against the algorithmic in + out + residual + weights of each convolution.
usage: layer_traffic.py <fetch counter_collection.csv> <write counter_collection.csv>"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_frame(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    start = max(i for i, r in enumerate(rows) if "nchw_to_nhwc" in r["Kernel_Name"])
    return [(r["Kernel_Name"], float(r["Counter_Value"])) for r in rows[start:]]


fetch = last_frame(sys.argv[1], "FETCH_SIZE")
write = last_frame(sys.argv[2], "WRITE_SIZE")
assert [k for k, _ in fetch] == [k for k, _ in write], "the two runs launched different kernel sequences"
layers = [r for r in json.load(open(os.path.join(ROOT, "gpurun_out", "layers.json"))) if r[1].startswith("conv_")]
# FETCH_SIZE: KB, reported at half the bytes of wide streaming reads on gfx950 (MI355X_MICROARCH.md) -> x2; WRITE_SIZE: KB
convs = []
pending = [0.0, 0.0]          # F(4x4) input transform: runs BEFORE the grouped GEMM of its layer
for (k, f), (_, w) in zip(fetch, write):
    if ("conv_splitk_reduce" in k or "wino4_output_kernel" in k) and convs:
        convs[-1][1] += 2 * f * 1024
        convs[-1][2] += w * 1024
        convs[-1][3] += 1
    elif "wino4_input_kernel" in k:
        pending[0] += 2 * f * 1024
        pending[1] += w * 1024
    elif "conv_igemm" in k or "conv_wino" in k:
        convs.append([k, 2 * f * 1024 + pending[0], w * 1024 + pending[1], 0])
        pending = [0.0, 0.0]
assert len(convs) == len(layers), (len(convs), len(layers))
tot_m = tot_a = 0
print(f"{'layer':70s} {'fetch MB':>9} {'write MB':>9} {'algor. MB':>9} {'ratio':>6}")
for (i, name, flops, us), (k, f, w, nred) in zip(layers, convs):
    m = re.search(r"\|(\d+)x(\d+)x(\d+)x(\d+)->(\d+) k(-?\d+) s(\d+) d(\d+) splitk(\d+)", name)
    alg = 0.0
    if m:
        B, H, W, cin, cout, k_, s_, d_, sk = (int(v) for v in m.groups())
        if k_ < 0:      # transposed conv k = stride = -k_
            oh, ow, taps = H * -k_, W * -k_, k_ * k_
        else:
            oh, ow, taps = (H - 1) // s_ + 1, (W - 1) // s_ + 1, k_ * k_
        alg = 4.0 * (B * H * W * cin + B * oh * ow * cout + cout * cin * taps)
    tot_m += f + w
    tot_a += alg
    print(f"{name[:70]:70s} {f / 1e6:9.1f} {w / 1e6:9.1f} {alg / 1e6:9.1f} {(f + w) / alg if alg else 0:6.2f}")
print(f"all convolution layers: measured {tot_m / 1e6:.0f} MB, algorithmic (without residuals) {tot_a / 1e6:.0f} MB, ratio {tot_m / tot_a:.2f}")
