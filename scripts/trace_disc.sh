#!/bin/bash
# ordered kernel list of one discriminator forward+backward -> gpurun_out/disc_trace.txt
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_disc -- python $ROOT/scripts/trace_disc.py > $OUT/prof_disc.log 2>&1
cd $ROOT
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof_disc/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 3
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
with open("$OUT/disc_trace.txt", "w") as w:
    tot = 0
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        w.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:10.1f} {d:9.1f} us  grid {r['Grid_Size_X']:>8s}  {r['Kernel_Name'][:120]}\n")
    w.write(f"sum of kernel time {tot:.1f} us, span {(int(rows[-1]['End_Timestamp'])-t0)/1e3:.1f} us\n")
PY
rm -rf $OUT/prof_disc
