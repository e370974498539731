#!/bin/bash
# cfg-5 / cfg-3 bf16 with and without the 64x128 tile of the direct-weight kernel (fresh per-layer measurement each)
mkdir -p gpurun_out/r3y
export SGV3D_NO_TUNE_DB=1 SGV3D_TUNE_ROUNDS=8 SGV3D_TUNE_REPEATS=4
for nr in 0 1; do
for st in 3 1; do
SGV3D_DW_NARROW=$nr SGV3D_TUNE_CACHE=gpurun_out/r3y/tune_nr${nr}.json python3 bench.py --sub --config cfg5 --batch 1 --dtype bf16 --steps 30 --warmup 3 --streams $st --no-cpu-baseline --no-roofline > gpurun_out/r3y/cfg5_nr${nr}_st${st}.json 2> gpurun_out/r3y/cfg5_nr${nr}_st${st}.err
echo "cfg5 narrow=$nr streams=$st rc=$? $(python3 -c "import json; d=json.loads(open('gpurun_out/r3y/cfg5_nr${nr}_st${st}.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))")"
done
done
python3 - <<PY
import json
a = json.load(open("gpurun_out/r3y/tune_nr1.json"))
import collections
c = collections.Counter(v[0] for v in a.values() if isinstance(v, list))
print("tile histogram:", dict(c))
PY
