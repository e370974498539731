mkdir -p gpurun_out/r3y
export SGV3D_NO_TUNE_DB=1 SGV3D_TUNE_ROUNDS=8 SGV3D_TUNE_REPEATS=4
for mx in 768 100000; do
SGV3D_DW_DEEP_MAX_WGS=$mx SGV3D_TUNE_CACHE=gpurun_out/r3y/tune3_mx${mx}.json python3 bench.py --sub --config cfg3 --batch 4 --dtype bf16 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r3y/cfg3_mx${mx}.json 2> gpurun_out/r3y/cfg3_mx${mx}.err
echo "cfg3 max=$mx rc=$? $(python3 -c "import json; d=json.loads(open('gpurun_out/r3y/cfg3_mx${mx}.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))")"
SGV3D_DW_DEEP_MAX_WGS=$mx SGV3D_TUNE_CACHE=gpurun_out/r3y/tune5b4_mx${mx}.json python3 bench.py --sub --config cfg5 --batch 4 --dtype bf16 --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r3y/cfg5b4_mx${mx}.json 2> gpurun_out/r3y/cfg5b4_mx${mx}.err
echo "cfg5b4 max=$mx rc=$? $(python3 -c "import json; d=json.loads(open('gpurun_out/r3y/cfg5b4_mx${mx}.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))")"
done
python3 -c "
import json,collections
a=json.load(open('gpurun_out/r3y/tune3_mx100000.json')); print(dict(sorted(collections.Counter(v[0] for v in a.values() if isinstance(v,list)).items())))"
