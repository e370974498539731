#!/bin/bash
# every kernel of the cfg-2 training step (torch glue included): rocprofv3 kernel trace of tools/train_bench.py + torch profiler table
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/train_trace
mkdir -p $OUT
cd $R
python3 tools/train_torch_profile.py > $OUT/torch_profile.txt 2> $OUT/torch_profile.err
echo "torch profile rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o train -- python3 $R/tools/train_bench.py --batch 2 --steps 5 --warmup 2 > $OUT/train_under_rocprof.json 2> $OUT/rocprof.err
echo "rocprof rc=$?"
tail -1 $OUT/train_under_rocprof.json | cut -c1-400
