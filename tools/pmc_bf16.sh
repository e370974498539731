#!/bin/bash
# PMC probe of the bf16 implicit-GEMM kernel on two representative shapes (tools/conv_modes.py, bf16 only):
#   bash tools/pmc_bf16.sh   ->  gpurun_out/pmc_bf16/summary.txt
R=$PWD; OUT=$PWD/gpurun_out/pmc_bf16; rm -rf $OUT; mkdir -p $OUT
export MODES=bf16 SHAPES="1,512,54,96,512,3,1,6,6;4,64,512,512,2304,3,1,1,1;1,64,216,384,256,1,1,0,1"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT -o a -- python3 $R/tools/conv_modes.py > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT -o b -- python3 $R/tools/conv_modes.py > /dev/null 2> $OUT/b.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT -o c -- python3 $R/tools/conv_modes.py > /dev/null 2> $OUT/c.err
cd $R; python tools/pmc_summary.py $OUT conv_igemm_bf16 > $OUT/summary.txt; tail -3 $OUT/c.err
