#!/bin/bash
# Re-measure only the layers whose committed choice is a three-launch F(4x4) tile (now that the f32x3 position GEMM, tiles 50-59,
# competes) and write the merged DB to gpurun_out/tune/gfx950_cfg2.json:   bash tools/retune_wino4.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tune
mkdir -p $OUT
cd $R
python3 - <<PY
import json
d = json.load(open("$R/tune/gfx950_cfg2.json"))
keep = {k: v for k, v in d.items() if v[0] not in (9, 10, 15, 46, 47) and not 50 <= v[0] < 60}
json.dump(keep, open("$OUT/gfx950_cfg2.json", "w"), indent=0, sort_keys=True)
print(len(d), "->", len(keep), "kept;", len(d) - len(keep), "to re-measure")
PY
export SGV3D_TUNE_SKIP=gfx950_cfg2.json
export SGV3D_TUNE_ROUNDS=8 SGV3D_TUNE_REPEATS=4
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg2.json python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs > $OUT/cfg2_x3.json 2> $OUT/cfg2_x3.err
echo "rc=$?"
python3 - <<PY
import json, collections
d = json.load(open("$OUT/gfx950_cfg2.json"))
print(len(d), sorted(collections.Counter(v[0] for v in d.values()).items()))
r = json.loads(open("$OUT/cfg2_x3.json").read().strip().splitlines()[-1])
print({k: r.get(k) for k in ("value", "ms_per_step")}, {k: v for k, v in r.items() if "one_frame" in k or "harness" in k and not isinstance(v, dict)})
PY
tail -3 $OUT/cfg2_x3.err
