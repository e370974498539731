mkdir -p gpurun_out/r3v
for st in 3 4 6 8; do
SGV3D_TUNE_STREAMS=3 python3 bench.py --sub --config cfg5 --batch 1 --dtype bf16 --steps 40 --warmup 5 --streams $st --no-cpu-baseline --no-roofline > gpurun_out/r3v/cfg5_st$st.json 2> gpurun_out/r3v/cfg5_st$st.err
echo "cfg5 streams=$st rc=$? $(python3 -c "import json; d=json.loads(open('gpurun_out/r3v/cfg5_st$st.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))")"
done
for st in 3 5; do
SGV3D_TUNE_STREAMS=3 python3 bench.py --sub --config cfg3 --batch 4 --dtype bf16 --steps 12 --warmup 3 --streams $st --no-cpu-baseline --no-roofline > gpurun_out/r3v/cfg3_st$st.json 2> gpurun_out/r3v/cfg3_st$st.err
echo "cfg3 streams=$st rc=$? $(python3 -c "import json; d=json.loads(open('gpurun_out/r3v/cfg3_st$st.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))")"
done
