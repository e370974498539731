#!/bin/bash
# rocprof kernel summary of the cfg-2 training step (tools/train_bench.py), copied to profiles/ afterwards
set -e
TAG=${1:-r03}
OUT=$PWD/gpurun_out/profiles_${TAG}_train
mkdir -p $OUT
export SGV3D_TUNE_CACHE=$OUT/train_tune_cache.json   # the profiled run replays the choices of the first run (training shapes are not in tune/)
python tools/train_bench.py --batch 2 --steps 5 --warmup 2 --profile > $OUT/train_bench.json 2> $OUT/train_bench.err
python tools/train_bench.py --batch 4 --steps 5 --warmup 2 > $OUT/train_bench_b4.json 2> $OUT/train_bench_b4.err
python tools/train_bench.py --config cfg5 --batch 2 --steps 3 --warmup 2 --profile > $OUT/train_bench_cfg5_b2.json 2> $OUT/train_bench_cfg5_b2.err || true
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof -o train -- python3 $REPO/tools/train_bench.py --batch 2 --steps 3 --warmup 1 > $OUT/train_under_rocprof.json 2> $OUT/rocprof.err || true
find $OUT -name "*kernel_stats.csv" | head
