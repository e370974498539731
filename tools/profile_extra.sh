#!/bin/bash
# bf16-config artefacts of the round (run from the repo root through gpurun):  bash tools/profile_extra.sh r02
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 $R/bench.py --config cfg3 --batch 4 --dtype bf16 --steps 10 --warmup 2 > $OUT/${TAG}_bench_cfg3_bf16_b4.json 2> $OUT/bench_cfg3.err
python3 $R/bench.py --config cfg5 --dtype bf16 --steps 20 --warmup 2 > $OUT/${TAG}_bench_cfg5_bf16.json 2> $OUT/bench_cfg5.err
python3 $R/bench.py --config cfg5 --batch 4 --dtype bf16 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_cfg5_bf16_b4.json 2> $OUT/bench_cfg5b4.err
python3 $R/bench.py --config cfg3 --batch 4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_cfg3_fp32_b4.json 2> $OUT/bench_cfg3f.err
python3 $R/bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_cfg5.json 2> $OUT/bench_cfg5f.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_bench_cfg3_bf16 -- python3 $R/bench.py --config cfg3 --batch 4 --dtype bf16 --steps 10 --warmup 2 --no-cpu-baseline --streams 1 > $OUT/${TAG}_bench_cfg3_bf16_under_rocprof.json 2> $OUT/rocprof_cfg3.err
ls -la $OUT | head -40
