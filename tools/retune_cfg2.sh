#!/bin/bash
# cfg-2 fp32 part of tools/make_tune_db.sh alone (three and one frame in flight, the harness's batch 8) -> gpurun_out/tune/gfx950_cfg2.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tune
mkdir -p $OUT
export SGV3D_NO_TUNE_DB=1 SGV3D_TUNE_ROUNDS=8 SGV3D_TUNE_REPEATS=4
cd $R
rm -f $OUT/gfx950_cfg2.json
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg2.json python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs --no-native-f32 --no-train-step > $OUT/cfg2.json 2> $OUT/cfg2.err
echo "cfg2 rc=$?"
python3 - <<PY
import json, collections
d = json.load(open("$OUT/gfx950_cfg2.json"))
print(len(d), sorted(collections.Counter(v[0] for v in d.values()).items()))
r = json.loads(open("$OUT/cfg2.json").read().strip().splitlines()[-1])
print({k: r.get(k) for k in ("value", "ms_per_step")})
PY
