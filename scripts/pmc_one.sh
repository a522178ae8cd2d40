#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel for one command (two separate --pmc passes).  usage: scripts/pmc_one.sh <tag> <python script> [env...]
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f_$TAG -- python "$@" > $OUT/pmc_f_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w_$TAG -- python "$@" > $OUT/pmc_w_$TAG.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
def agg(d, c):
    f = glob.glob(f"$OUT/{d}/**/*counter_collection.csv", recursive=True)
    a = collections.defaultdict(lambda: [0.0, 0, 0.0]); seen = set()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"][:60]; a[k][0] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); a[k][1] += 1; a[k][2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return a
f, w = agg("pmc_f_$TAG", "FETCH_SIZE"), agg("pmc_w_$TAG", "WRITE_SIZE")
for k in sorted(f, key=lambda k: -f[k][2])[:8]:
    n = f[k][1]
    print(f"{k:60s} n={n:4d} avg {f[k][2]/n/1e3:8.1f} us  fetch(x2) {2*f[k][0]*1024/n/1e6:8.1f} MB  write {w[k][0]*1024/max(1,w[k][1])/1e6:8.1f} MB")
PY
rm -rf $OUT/pmc_f_$TAG $OUT/pmc_w_$TAG
