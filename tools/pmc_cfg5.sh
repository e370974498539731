#!/bin/bash
# issue-slot accounting of a cfg-5 bf16 frame: MFMA and VALU instruction counts per kernel (one PMC pass, one frame in flight)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_cfg5
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT -o cfg5 -- python3 $R/bench.py --sub --config cfg5 --batch 1 --dtype bf16 --steps 10 --warmup 2 --streams 1 --no-cpu-baseline --no-roofline > $OUT/run.json 2> $OUT/run.err
echo "rc=$?"
ls $OUT | head
