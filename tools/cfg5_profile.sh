#!/bin/bash
# per-kernel time of one cfg-5 bf16 frame (batch 1): rocprofv3 kernel trace of bench.py, one frame in flight and three
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cfg5prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for st in 1 3; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o cfg5_bf16_st$st -- python3 $R/bench.py --sub --config cfg5 --batch 1 --dtype bf16 --steps 30 --warmup 3 --streams $st --no-cpu-baseline --no-roofline > $OUT/cfg5_st$st.json 2> $OUT/cfg5_st$st.err
echo "streams=$st rc=$?"
done
ls $OUT
