mkdir -p gpurun_out/r3w
run() { # name args...
  n=$1; shift
  python3 bench.py --sub "$@" --no-cpu-baseline --no-roofline > gpurun_out/r3w/$n.json 2> gpurun_out/r3w/$n.err
  echo "$n rc=$? $(python3 -c "import json; d=json.loads(open('gpurun_out/r3w/$n.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('parity',{}).get('max_abs_err'))")"
}
run cfg5_plain --config cfg5 --batch 1 --dtype bf16 --steps 40 --warmup 5 --no-fuse-lift-splat
run cfg5_fused --config cfg5 --batch 1 --dtype bf16 --steps 40 --warmup 5
run cfg3_plain --config cfg3 --batch 4 --dtype bf16 --steps 12 --warmup 3 --no-fuse-lift-splat
run cfg3_fused --config cfg3 --batch 4 --dtype bf16 --steps 12 --warmup 3
run cfg2_plain --steps 40 --warmup 5 --no-fuse-lift-splat
run cfg2_fused --steps 40 --warmup 5
