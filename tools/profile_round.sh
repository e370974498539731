#!/bin/bash
# Produce the round's profile artefacts on the GPU box (run from the repo root through gpurun):
#   bash tools/profile_round.sh r01
# Outputs land in gpurun_out/profiles_<tag>/ ; copy what should be judged into profiles/.
set -u
# (PARTS=a and PARTS=b on different boxes: profiles/<tag>_hbm_traffic.json is the two calls' sections put together by hand)
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
# raw rocprofv3 output (kernel traces, counter collections: > 64 MiB, more than gpurun copies back) stays on the box under /tmp;
# the small summaries tools/summarize_profiles.py makes of it go to gpurun_out/profiles_<tag>/ and from there, by hand, to profiles/
OUT=/tmp/sgv3d_profiles_$TAG
KEEP=$R/gpurun_out/profiles_$TAG
PARTS=${PARTS:-abc}     # a = bench line, kernel traces, PMC passes (steps 1-3b); b = voxel pooling, harness, gather probe (4-6); c = bf16 configs (7); the
                        # summaries are made at the end of every call from whatever raw files are there (gpurun calls are <= 20 min)
if [[ $PARTS == *a* ]]; then rm -rf $OUT; fi
mkdir -p $OUT $KEEP
if [[ $PARTS == *a* ]]; then
export TMPDIR=/tmp
cd /tmp
# (tile / split-K choices come from the committed tune DB, tune/gfx950_*.json: every run below makes the same ones)
# (1. the bench line itself runs LAST, below: it pairs its flops per launch with this round's roofline trace and PMC summary)
# 2. kernel trace + stats of the same command (no CPU baseline: it is host work).  One frame in flight, so that the
echo "[profile_round] $(date +%H:%M:%S) 2. kernel trace + stats of the same command (no CPU baseli" | tee -a $KEEP/progress.log
#    per-kernel durations are those of the kernels alone (with 3 frames in flight concurrent kernels stretch each
#    other) and compare directly with roofline.avg_launch_us of the bench line, which is measured the same way.
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_bench -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-plan-timing --no-other-configs --no-harness --no-train-step --no-native-f32 --streams 1 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/rocprof.err
# 2r. the ROOFLINE population alone: bench.py --roofline-only = warm-up + the instrumented pass, one stream, eager launches from the
#     first to the last forward -- the per-symbol averages of this trace are over exactly the launches roofline.frac is made of
#     (bench.py pairs its own flops per launch with this file: roofline.frac_from_rocprof)
echo "[profile_round] $(date +%H:%M:%S) 2r. bench.py --roofline-only under the kernel trace" | tee -a $KEEP/progress.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_bench_roofline -- python3 $R/bench.py --roofline-only --steps 20 --warmup 3 > $OUT/${TAG}_bench_roofline_under_rocprof.json 2> $OUT/rocprof_roofline.err
# 2b. the HEADLINE command itself (three frames in flight) under the kernel trace: concurrent kernels stretch each other,
echo "[profile_round] $(date +%H:%M:%S) 2b. the HEADLINE command itself (three frames in flight) u" | tee -a $KEEP/progress.log
#     so these per-kernel durations are "under load" figures, not comparable with roofline.avg_launch_us
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_bench_3inflight -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs --no-harness > $OUT/${TAG}_bench_3inflight_under_rocprof.json 2> $OUT/rocprof3.err
# 2c. per-layer table (HIP events, one frame at a time) with the tiles chosen for three frames in flight, and for one
echo "[profile_round] $(date +%H:%M:%S) 2c. per-layer table (HIP events, one frame at a time) with" | tee -a $KEEP/progress.log
SGV3D_TUNE_STREAMS=3 python3 $R/tools/layer_report.py > $OUT/${TAG}_layers_cfg2_tiles_for_3inflight.txt 2> $OUT/layers3.err
SGV3D_TUNE_STREAMS=1 python3 $R/tools/layer_report.py > $OUT/${TAG}_layers_cfg2_tiles_for_1inflight.txt 2> $OUT/layers1.err
# 2d. the three launches of the F(4x4) Winograd path
echo "[profile_round] $(date +%H:%M:%S) 2d. the three launches of the F(4x4) Winograd path" | tee -a $KEEP/progress.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_wino4 -- python3 $R/tools/wino4_trace.py > /dev/null 2> $OUT/wino4.err
# 3. HBM traffic counters, separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
echo "[profile_round] $(date +%H:%M:%S) 3. HBM traffic counters, separate passes (FETCH_SIZE and W" | tee -a $KEEP/progress.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o ${TAG}_pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs --no-harness > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o ${TAG}_pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs --no-harness > /dev/null 2> $OUT/pmc_write.err
# 3b. MFMA utilisation counters (own pass; SQ counters fit one pass)
echo "[profile_round] $(date +%H:%M:%S) 3b. MFMA utilisation counters (own pass; SQ counters fit o" | tee -a $KEEP/progress.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o ${TAG}_pmc_mfma -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs --no-harness --streams 1 > /dev/null 2> $OUT/pmc_mfma.err
fi
if [[ $PARTS == *c* ]]; then
# 7. the bf16 configurations: one frame in flight under the kernel trace (cfg-3 batch 4, cfg-5 batch 1), and their per-layer tables
echo "[profile_round] $(date +%H:%M:%S) 7. bf16 cfg-3 / cfg-5 kernel traces and layer tables" | tee -a $KEEP/progress.log
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_bench_cfg3_bf16 -- python3 $R/bench.py --sub --config cfg3 --batch 4 --dtype bf16 --steps 12 --warmup 3 --streams 1 --no-cpu-baseline --no-roofline > $OUT/${TAG}_bench_cfg3_bf16_under_rocprof.json 2> $OUT/cfg3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_bench_cfg5_bf16 -- python3 $R/bench.py --sub --config cfg5 --batch 1 --dtype bf16 --steps 30 --warmup 3 --streams 1 --no-cpu-baseline --no-roofline > $OUT/${TAG}_bench_cfg5_bf16_under_rocprof.json 2> $OUT/cfg5.err
CONFIG=cfg3 DTYPE=bf16 BATCH=4 python3 $R/tools/layer_report.py > $OUT/${TAG}_layers_cfg3_bf16_b4.txt 2> $OUT/layers_cfg3.err
CONFIG=cfg5 DTYPE=bf16 BATCH=1 python3 $R/tools/layer_report.py > $OUT/${TAG}_layers_cfg5_bf16_b1.txt 2> $OUT/layers_cfg5.err
fi
if [[ $PARTS == *b* ]]; then
# 4. voxel pooling micro-benchmark (the HBM-bound headline kernel) + its trace and traffic
echo "[profile_round] $(date +%H:%M:%S) 4. voxel pooling micro-benchmark (the HBM-bound headline k" | tee -a $KEEP/progress.log
python3 $R/tools/microbench.py --what vp,lift --out $OUT/${TAG}_voxel_pooling_microbench.json > $OUT/microbench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_vp -- python3 $R/tools/vp_probe.py > /dev/null 2> $OUT/vp.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o ${TAG}_vp_pmc_fetch -- python3 $R/tools/vp_probe.py > /dev/null 2>> $OUT/vp.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o ${TAG}_vp_pmc_write -- python3 $R/tools/vp_probe.py > /dev/null 2>> $OUT/vp.err
# 5. the reference harness's eval_step (sgv3d_amd/harness.py): phase timing, and the kernel trace of the same loop
echo "[profile_round] $(date +%H:%M:%S) 5. the reference harness's eval_step (sgv3d_amd/harness.py" | tee -a $KEEP/progress.log
python3 $R/tools/harness_profile.py > $OUT/${TAG}_harness_phases.txt 2> $OUT/harness.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_harness -- python3 $R/tools/harness_profile.py > /dev/null 2>> $OUT/harness.err
# 6. the gather kernels side by side (voxel-owner / slot-balanced) on the cfg-2 / cfg-5 / cfg-3 geometries
echo "[profile_round] $(date +%H:%M:%S) 6. the gather kernels side by side (voxel-owner / slot-bal" | tee -a $KEEP/progress.log
python3 $R/tools/vp_probe3.py > $OUT/${TAG}_gather_probe.txt 2> $OUT/gather_probe.err
SGV3D_VP_KERNEL=slot python3 $R/tools/vp_probe3.py >> $OUT/${TAG}_gather_probe.txt 2>> $OUT/gather_probe.err
fi
ls -la $OUT | head -60
python3 $R/tools/summarize_profiles.py $OUT $TAG $KEEP > $KEEP/summarize.log 2>&1
cp $OUT/*.err $KEEP/ 2>/dev/null
if [[ $PARTS == *a* ]]; then
# 1. the bench line itself (with the CPU baseline), after this round's roofline trace + the line printed under it and the PMC summary
#    have been put where bench.py reads them (profiles/ of this checkout): roofline.frac_from_rocprof and roofline.traffic then come
#    from the SAME tree's traces
cp $KEEP/${TAG}_bench_roofline_kernel_stats.csv $KEEP/${TAG}_bench_roofline_under_rocprof.json $KEEP/${TAG}_hbm_traffic.json $R/profiles/
echo "[profile_round] $(date +%H:%M:%S) 1. the bench line itself (with the CPU baseline)" | tee -a $KEEP/progress.log
python3 $R/bench.py --steps 20 --warmup 3 > $KEEP/${TAG}_bench.json 2> $KEEP/bench.err
fi
ls -la $KEEP
