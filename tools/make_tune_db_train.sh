#!/bin/bash
# The training part of tools/make_tune_db.sh alone: forward / data-gradient / weight-gradient choices of a cfg-2 step at batch 2
# and 4 -> gpurun_out/tune/gfx950_cfg2_train.json (only the signatures the inference DBs do not hold)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tune
mkdir -p $OUT
export SGV3D_TUNE_ROUNDS=8 SGV3D_TUNE_REPEATS=4
cd $R
rm -f $OUT/gfx950_cfg2_train.json
# measure the training signatures afresh, keep the inference DBs: the committed training DB is SKIPPED by the loader (never moved
# out of the tree: a killed run used to leave the working copy without it)
export SGV3D_TUNE_SKIP=gfx950_cfg2_train.json
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg2_train.json python3 tools/train_bench.py --batch 2 --steps 3 > $OUT/train_b2.json 2> $OUT/train_b2.err
echo "train b2 rc=$?"
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg2_train.json python3 tools/train_bench.py --batch 4 --steps 3 > $OUT/train_b4.json 2> $OUT/train_b4.err
echo "train b4 rc=$?"
# ... and the mixed-precision step (bf16 products: "|bf16" forward / data-gradient signatures, "wgrad|...xbf16" weight gradients)
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg2_train.json python3 tools/train_bench.py --batch 2 --steps 3 --dtype bf16 > $OUT/train_b2_bf16.json 2> $OUT/train_b2_bf16.err
echo "train b2 bf16 rc=$?"
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg2_train.json python3 tools/train_bench.py --batch 4 --steps 3 --dtype bf16 > $OUT/train_b4_bf16.json 2> $OUT/train_b4_bf16.err
echo "train b4 bf16 rc=$?"
unset SGV3D_TUNE_SKIP
python3 - <<PY
import json, glob
t = json.load(open("$OUT/gfx950_cfg2_train.json"))
o = {}
for f in glob.glob("$R/tune/gfx950_*.json"):
    if "train" not in f:
        o.update(json.load(open(f)))
json.dump({k: v for k, v in t.items() if k not in o}, open("$OUT/gfx950_cfg2_train.json", "w"), indent=0, sort_keys=True)
print(len(t), "->", len([k for k in t if k not in o]))
PY
tail -c 400 $OUT/train_b2.json; echo; tail -c 400 $OUT/train_b4.json
