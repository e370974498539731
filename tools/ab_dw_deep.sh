#!/bin/bash
# cfg-5 / cfg-3 bf16 with and without the two-chunks-ahead tiles of the direct-weight kernel (fresh per-layer measurement each)
mkdir -p gpurun_out/r3x
export SGV3D_NO_TUNE_DB=1 SGV3D_TUNE_ROUNDS=8 SGV3D_TUNE_REPEATS=4
for dp in 0 1; do  # (SGV3D_DW_DEEP also gates the 64x128 tile, see hip_ops)
for st in 3 1; do
SGV3D_DW_DEEP=$dp SGV3D_TUNE_CACHE=gpurun_out/r3x/tune_dp${dp}.json python3 bench.py --sub --config cfg5 --batch 1 --dtype bf16 --steps 30 --warmup 3 --streams $st --no-cpu-baseline --no-roofline > gpurun_out/r3x/cfg5_dp${dp}_st${st}.json 2> gpurun_out/r3x/cfg5_dp${dp}_st${st}.err
echo "cfg5 deep=$dp streams=$st rc=$? $(python3 -c "import json; d=json.loads(open('gpurun_out/r3x/cfg5_dp${dp}_st${st}.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))")"
done
SGV3D_DW_DEEP=$dp SGV3D_TUNE_CACHE=gpurun_out/r3x/tune3_dp${dp}.json python3 bench.py --sub --config cfg3 --batch 4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r3x/cfg3_dp${dp}.json 2> gpurun_out/r3x/cfg3_dp${dp}.err
echo "cfg3 deep=$dp rc=$? $(python3 -c "import json; d=json.loads(open('gpurun_out/r3x/cfg3_dp${dp}.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))")"
done
python3 - <<PY
import json
for f in ("gpurun_out/r3x/tune_dp1.json", "gpurun_out/r3x/tune3_dp1.json"):
    a = json.load(open(f))
    n = sum(1 for v in a.values() if isinstance(v, list) and v[0] in (36, 37))
    print(f, "signatures on a DEEP tile:", n, "of", len(a))
PY
