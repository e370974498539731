#!/bin/bash
# training step, batch 2: with and without the gradient slots (one box, back to back)
mkdir -p gpurun_out/r3t
for dg in 1 0 1; do
SGV3D_DIRECT_GRADS=$dg python3 tools/train_bench.py --batch 2 --steps 8 --warmup 3 > gpurun_out/r3t/train_dg$dg.json 2> gpurun_out/r3t/train_dg$dg.err
echo "direct_grads=$dg rc=$? $(python3 -c "import json; d=json.loads(open('gpurun_out/r3t/train_dg$dg.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('loss'))")"
done
