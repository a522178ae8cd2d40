#!/bin/bash
# kernel stats of bench.py at another per-GPU batch (which kernels carry per-launch fixed cost): gpurun_out/kstats_b<batch>.txt
set -e
B=${1:-4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b$B -- python $ROOT/bench.py --batch $B --no-cpu-baseline --no-kernel-profile > $OUT/prof_b$B.log 2>&1
cd $ROOT
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof_b$B/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("$OUT/kstats_b$B.txt", "w") as w:
    for r in rows[:60]:
        w.write(f"{r['Name'][:110]:110s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.3f} {float(r['AverageNs'])/1e3:9.2f}\n")
PY
rm -rf $OUT/prof_b$B
