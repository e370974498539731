#!/bin/bash
# PMC probe of the implicit-GEMM f32x3 kernel (conv_pw_x3_kernel) on one layer shape:  bash tools/pmc_pw_x3.sh  ->  gpurun_out/pmc_pw_x3/summary.txt
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/pmc_pw_x3; rm -rf $OUT; mkdir -p $OUT
export ONLY="${ONLY:-512->512 @54x96}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT -o a -- python3 $R/tools/pw_x3_probe.py > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT -o b -- python3 $R/tools/pw_x3_probe.py > /dev/null 2> $OUT/b.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT -o c -- python3 $R/tools/pw_x3_probe.py > /dev/null 2> $OUT/c.err
cd $R; python3 tools/pmc_summary.py $OUT conv_pw_x3_kernel > $OUT/summary.txt; grep -A6 "kernel<2,4,0>\|kernel<2, 4, 0>" $OUT/summary.txt | head -60
