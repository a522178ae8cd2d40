#!/bin/bash
# Round profile: run on the GPU box through gpurun from the repo root.
#   1. bench.py (default arguments) -> gpurun_out/bench_rNN.json
#   2. rocprofv3 --kernel-trace --stats of the same bench command (CPU baseline leg skipped: identical GPU work)
#   3. HBM traffic of the same command: FETCH_SIZE and WRITE_SIZE in two separate --pmc passes
#      (MI355X_MICROARCH.md: they do not fit one pass; no trace domains combined with --pmc)
# The summaries are condensed into profiles/ by scripts/summarize_profile.py (run in the container).
#   scripts/profile_round.sh r03            the headline configuration -> <R>_kernel_stats.txt, <R>_hbm_traffic.json
#   scripts/profile_round.sh r03 cycle      another bench configuration -> <R>_cycle_kernel_stats.txt, <R>_cycle_hbm_traffic.json (bench.py
#                                           reads the traffic file of ITS configuration: roofline.traffic is then not null for it either)
set -e
R=${1:-r01}
CFG=${2:-paired}
ARGS=""
BASE=""
if [ "$CFG" != "paired" ]; then ARGS="--config $CFG --steps 3 --warmup 1"; BASE="--no-cpu-baseline"; R=${R}_$CFG; fi    # (the CPU leg of the big configurations runs for minutes: scripts/bench_configs.sh has it)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
# heartbeat: the counter passes of the big configurations write nothing for minutes (a silent GPU command is taken to be hung)
( while true; do sleep 60; echo "[profile_round] $R: still running $(date +%T)"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
echo "[profile_round] $R: bench"; timeout -k 10 900 python bench.py $ARGS $BASE > $OUT/bench_$R.json 2> $OUT/bench_$R.err
cd /tmp && export TMPDIR=/tmp
echo "[profile_round] $R: kernel trace"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$R -- python $ROOT/bench.py --no-cpu-baseline $ARGS > $OUT/prof_$R.log 2>&1
echo "[profile_round] $R: FETCH_SIZE pass"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$R -- python $ROOT/bench.py --no-cpu-baseline --no-kernel-profile $ARGS --steps 2 --warmup 1 > $OUT/pmc_fetch_$R.log 2>&1
echo "[profile_round] $R: WRITE_SIZE pass"
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$R -- python $ROOT/bench.py --no-cpu-baseline --no-kernel-profile $ARGS --steps 2 --warmup 1 > $OUT/pmc_write_$R.log 2>&1
# keep only the small per-kernel aggregates (the raw traces exceed the 64 MiB merge cap)
cd $ROOT
python scripts/summarize_profile.py $R
rm -rf $OUT/prof_$R/*/*kernel_trace.csv $OUT/pmc_fetch_$R $OUT/pmc_write_$R
cat $OUT/bench_$R.json
