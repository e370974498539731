#!/bin/bash
# The cfg-2 fp32 part of tools/make_tune_db.sh alone (three frames in flight "|ts3" and one "|ts1" in the same run), plus the
# batch-8 harness step's signatures: -> gpurun_out/tune/gfx950_cfg2.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tune
mkdir -p $OUT
export SGV3D_NO_TUNE_DB=1          # measure everything afresh
export SGV3D_TUNE_ROUNDS=8 SGV3D_TUNE_REPEATS=4     # careful timing: near-ties are not to be decided by noise
cd $R
rm -f $OUT/gfx950_cfg2.json
SGV3D_TUNE_CACHE=$OUT/gfx950_cfg2.json python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs > $OUT/cfg2.json 2> $OUT/cfg2.err
echo "cfg2 rc=$?"
wc -c $OUT/gfx950_cfg2.json
