#!/bin/bash
# Counters of the all-taps bf16 weight gradient (conv_wgrad3x3_bf16_kernel): MFMA busy, LDS bank conflicts, waits.  Separate --pmc passes
# with --kernel-trace only.  bash tools/pmc_wgrad3x3_bf16.sh  (through gpurun); summary -> gpurun_out/pmc_wgrad3x3/summary.txt
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/pmc_wgrad3x3; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o a -- python3 $R/tools/wgrad3x3_bf16_pmc_driver.py > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT -o b -- python3 $R/tools/wgrad3x3_bf16_pmc_driver.py > /dev/null 2> $OUT/b.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT -o c -- python3 $R/tools/wgrad3x3_bf16_pmc_driver.py > /dev/null 2> $OUT/c.err
cd $R
for f in $(find $OUT -name "*counter_collection.csv"); do python3 tools/parse_pmc.py $f wgrad3x3_bf16; done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt; tail -2 $OUT/b.err
