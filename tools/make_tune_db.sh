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
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg2.json python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs --no-native-f32 --no-train-step > $OUT/cfg2.json 2> $OUT/cfg2.err
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
# training step (cfg-2 model at batch 2 and at cfg-4's per-GPU batch 4): forward / data-gradient choices and the weight-gradient
# (tile, split) choices ("wgrad|" signatures) -> gfx950_cfg2_train.json
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg2_train.json python3 tools/train_bench.py --batch 2 --steps 3 > $OUT/train_b2.json 2> $OUT/train_b2.err
echo "train b2 rc=$?"
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg2_train.json python3 tools/train_bench.py --batch 4 --steps 3 > $OUT/train_b4.json 2> $OUT/train_b4.err
echo "train b4 rc=$?"
wc -c $OUT/gfx950_*.json
# (the file written by train_bench.py holds the whole in-memory table: keep only the signatures that are not in the other files)
python3 - <<PY
import json, glob
t = json.load(open("$OUT/gfx950_cfg2_train.json"))
o = {}
for f in glob.glob("$OUT/gfx950_*.json"):
    if "train" not in f:
        o.update(json.load(open(f)))
json.dump({k: v for k, v in t.items() if k not in o}, open("$OUT/gfx950_cfg2_train.json", "w"), indent=0, sort_keys=True)
PY
