R=$PWD; OUT=$PWD/gpurun_out/pmc_wgrad; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT -o a -- python3 $R/tools/wgrad_probe.py > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT -o b -- python3 $R/tools/wgrad_probe.py > /dev/null 2> $OUT/b.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT -o c -- python3 $R/tools/wgrad_probe.py > /dev/null 2> $OUT/c.err
cd $R; python tools/pmc_summary.py $OUT wgrad_kernel; tail -2 $OUT/b.err
