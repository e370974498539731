#!/bin/bash
# Profiles of the cfg-2 training step (tools/train_bench.py), f32 and mixed precision (--dtype bf16): bench lines with per-kernel-family
# times, per-layer tables, rocprofv3 kernel summaries.  Run through gpurun; copy what should be judged from
# gpurun_out/profiles_<tag>_train/ to profiles/.      bash tools/profile_train.sh r05
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_${TAG}_train
mkdir -p $OUT
cd $R
# (per-layer choices come from the committed tune/gfx950_cfg2_train.json: every run below makes the same ones)
for dt in f32 bf16; do
  for b in 2 4; do
    python3 tools/train_bench.py --batch $b --steps 5 --warmup 3 --dtype $dt --profile > $OUT/${TAG}_train_bench_b${b}_${dt}.json 2> $OUT/train_b${b}_${dt}.err
    echo "train b$b $dt rc=$? $(python3 -c "import json; d=json.loads(open('$OUT/${TAG}_train_bench_b${b}_${dt}.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms')")"
  done
  DTYPE=$dt TOP=60 python3 tools/train_layer_report.py > $OUT/${TAG}_train_layers_b2_${dt}.txt 2> $OUT/layers_${dt}.err
done
# the step as one hipGraph replay (train_step.GraphedTrainStep): host-independent step time
for b in 2 4; do
  python3 tools/train_bench.py --batch $b --steps 10 --warmup 3 --dtype bf16 --graph > $OUT/${TAG}_train_bench_b${b}_bf16_graph.json 2> $OUT/train_b${b}_bf16_graph.err
  echo "train b$b bf16 graph rc=$? $(python3 -c "import json; d=json.loads(open('$OUT/${TAG}_train_bench_b${b}_bf16_graph.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms')")"
done
python3 tools/train_bench.py --batch 2 --steps 10 --warmup 3 --graph > $OUT/${TAG}_train_bench_b2_f32_graph.json 2> $OUT/train_b2_f32_graph.err
python3 tools/train_bench.py --config cfg5 --batch 2 --steps 3 --warmup 2 --profile > $OUT/${TAG}_train_bench_cfg5_b2_f32.json 2> $OUT/train_cfg5.err
echo "cfg5 b2 rc=$?"
python3 tools/train_bench.py --config cfg5 --batch 2 --steps 5 --warmup 3 --dtype bf16 --graph > $OUT/${TAG}_train_bench_cfg5_b2_bf16_graph.json 2> $OUT/train_cfg5_bf16_graph.err
echo "cfg5 b2 bf16 graph rc=$?"
python3 tools/bn_probe.py > $OUT/${TAG}_bn_probe.txt 2> $OUT/bn_probe.err
cd /tmp && export TMPDIR=/tmp
for dt in f32 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_$dt -o train -- python3 $R/tools/train_bench.py --batch 2 --steps 3 --warmup 2 --dtype $dt > $OUT/train_under_rocprof_$dt.json 2> $OUT/rocprof_$dt.err
  f=$(find $OUT/rocprof_$dt -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_train_kernel_stats_${dt}.csv
  rm -rf $OUT/rocprof_$dt
done
ls -la $OUT
