#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/ovl; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --output-format csv -d $OUT -o ovl -- python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs --no-harness --no-train-step --no-native-f32 > $OUT/line.json 2> $OUT/err.txt
python3 tools/overlap_report.py $(ls $OUT/*kernel_trace.csv | head -1) ${1:-60} ${2:-16} | tee $R/gpurun_out/overlap_report.txt
