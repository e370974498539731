#!/bin/bash
# per-kernel times of the F(4x4) path, f32 and f32x3 position GEMM, on one layer shape (ONLY="512->512 @54x96" by default)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/x3prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
export ONLY="${ONLY:-512->512 @54x96}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o x3 -- python3 $R/tools/wino4_x3_probe.py > $OUT/probe.log 2> $OUT/probe.err
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("gemm", "wino4", "igemm")):
        print(r["Name"][:100], r["Calls"], r["AverageNs"], r["MinNs"])
PY
tail -2 $OUT/probe.log
