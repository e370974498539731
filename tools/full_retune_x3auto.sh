R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tune_x3auto
mkdir -p $OUT
cd $R
export SGV3D_NO_TUNE_DB=1 SGV3D_TUNE_ROUNDS=8 SGV3D_TUNE_REPEATS=4 SGV3D_F32X3=auto
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg2.json python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs > $OUT/cfg2.json 2> $OUT/cfg2.err
echo "rc=$?"
python3 - <<PY
import json, collections
d = json.load(open("$OUT/gfx950_cfg2.json"))
print(len(d), sorted(collections.Counter(v[0] for v in d.values()).items()))
for k, v in sorted(d.items()):
    if 10 < v[0] < 20 and v[0] != 15: print(k, v)
r = json.loads(open("$OUT/cfg2.json").read().strip().splitlines()[-1])
print({k: r.get(k) for k in ("value", "ms_per_step")}, {k: v for k, v in r.items() if ("one_frame" in k or "harness" in k) and not isinstance(v, dict)})
PY
tail -3 $OUT/cfg2.err
