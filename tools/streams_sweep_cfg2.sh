#!/bin/bash
# cfg-2 fp32 frames/s against the number of frames in flight, one box, same committed per-layer choices ("|ts3")
mkdir -p gpurun_out/streams
for st in 1 2 3 4 5 6; do
SGV3D_TUNE_STREAMS=3 python3 bench.py --steps 60 --warmup 5 --streams $st --no-cpu-baseline --no-roofline --no-other-configs --no-harness > gpurun_out/streams/cfg2_st$st.json 2> gpurun_out/streams/cfg2_st$st.err
echo "cfg2 streams=$st rc=$? $(python3 -c "import json; d=json.loads(open('gpurun_out/streams/cfg2_st$st.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['long_run_value'] or 0,1) if 'long_run_value' in d else '')")"
done
