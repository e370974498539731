#!/bin/bash
# bf16 tune DBs only (cfg-3 / cfg-5)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tune
mkdir -p $OUT
export SGV3D_NO_TUNE_DB=1 SGV3D_TUNE_ROUNDS=8 SGV3D_TUNE_REPEATS=4
cd $R
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg3_bf16.json python3 bench.py --sub --config cfg3 --batch 4 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/cfg3.json 2> $OUT/cfg3.err
echo "cfg3 rc=$?"
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg5_bf16.json python3 bench.py --sub --config cfg5 --batch 1 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/cfg5.json 2> $OUT/cfg5.err
echo "cfg5 rc=$?"
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg5_bf16.json python3 bench.py --sub --config cfg5 --batch 4 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/cfg5b4.json 2> $OUT/cfg5b4.err
echo "cfg5 b4 rc=$?"
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg3_bf16.json python3 bench.py --sub --config cfg3 --batch 4 --dtype bf16 --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --no-roofline > $OUT/cfg3s1.json 2> $OUT/cfg3s1.err
echo "cfg3 streams 1 rc=$?"
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg5_bf16.json python3 bench.py --sub --config cfg5 --batch 1 --dtype bf16 --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --no-roofline > $OUT/cfg5s1.json 2> $OUT/cfg5s1.err
echo "cfg5 streams 1 rc=$?"
wc -c $OUT/gfx950_*.json
