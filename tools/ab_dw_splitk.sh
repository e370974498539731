mkdir -p gpurun_out/r3s
export SGV3D_NO_TUNE_DB=1 SGV3D_TUNE_ROUNDS=8 SGV3D_TUNE_REPEATS=4
for sk in 0 1; do
for st in 3 1; do
SGV3D_DW_SPLITK=$sk SGV3D_TUNE_CACHE=gpurun_out/r3s/tune_sk${sk}.json python3 bench.py --sub --config cfg5 --batch 1 --dtype bf16 --steps 20 --warmup 3 --streams $st --no-cpu-baseline --no-roofline > gpurun_out/r3s/cfg5_sk${sk}_st${st}.json 2> gpurun_out/r3s/cfg5_sk${sk}_st${st}.err
echo "sk=$sk streams=$st rc=$? $(python3 -c "import json,sys; d=json.loads(open('gpurun_out/r3s/cfg5_sk${sk}_st${st}.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done; done
grep -o '"[^"]*": \[3[1-5], [2-9]\]' gpurun_out/r3s/tune_sk1.json | head -40
