#!/bin/bash
# Measure the per-layer (algorithm, tile, split-K) choices of the BASELINE configurations on this MI355X and write them to
# gpurun_out/tune/ (copy to tune/gfx950_*.json to commit them; hip_ops loads tune/gfx950_*.json by default):
#   bash tools/make_tune_db.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tune
mkdir -p $OUT
export SGV3D_NO_TUNE_DB=1          # measure everything afresh
export SGV3D_TUNE_ROUNDS=8 SGV3D_TUNE_REPEATS=4     # careful timing: near-ties are not to be decided by noise
cd $R
# cfg-2 fp32: three frames in flight ("|ts3" signatures) and, in the same run, one frame in flight ("|ts1")
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg2.json python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs > $OUT/cfg2.json 2> $OUT/cfg2.err
echo "cfg2 rc=$?"
# cfg-3 / cfg-5 in bf16 at the batch sizes other_configs runs (and batch 4 for cfg-5)
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg3_bf16.json python3 bench.py --sub --config cfg3 --batch 4 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/cfg3.json 2> $OUT/cfg3.err
echo "cfg3 rc=$?"
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg5_bf16.json python3 bench.py --sub --config cfg5 --batch 1 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/cfg5.json 2> $OUT/cfg5.err
echo "cfg5 rc=$?"
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg5_bf16.json python3 bench.py --sub --config cfg5 --batch 4 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/cfg5b4.json 2> $OUT/cfg5b4.err
echo "cfg5 b4 rc=$?"
# one frame in flight for the bf16 configs too ("|ts1": what the --streams 1 profile runs replay)
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg3_bf16.json python3 bench.py --sub --config cfg3 --batch 4 --dtype bf16 --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --no-roofline > $OUT/cfg3s1.json 2> $OUT/cfg3s1.err
echo "cfg3 streams 1 rc=$?"
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg5_bf16.json python3 bench.py --sub --config cfg5 --batch 1 --dtype bf16 --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --no-roofline > $OUT/cfg5s1.json 2> $OUT/cfg5s1.err
echo "cfg5 streams 1 rc=$?"
wc -c $OUT/gfx950_*.json
