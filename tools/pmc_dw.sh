#!/bin/bash
# PMC probe (MFMA busy, LDS conflicts, wait states, VMEM) of the bf16 direct-weight kernel:  bash tools/pmc_dw.sh -> gpurun_out/pmc_dw/summary.txt
R=$PWD; OUT=$PWD/gpurun_out/pmc_dw; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o a -- python3 $R/tools/dw_pmc_probe.py > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o b -- python3 $R/tools/dw_pmc_probe.py > /dev/null 2> $OUT/b.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o c -- python3 $R/tools/dw_pmc_probe.py > /dev/null 2> $OUT/c.err
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o d -- python3 $R/tools/dw_pmc_probe.py > /dev/null 2> $OUT/d.err
cd $R
for f in a b c d; do for c in $(find $OUT -name "${f}_counter_collection.csv"); do echo "== pass $f"; python tools/parse_pmc.py $c conv_dw; done; done > $OUT/summary.txt 2>&1
tail -3 $OUT/d.err
