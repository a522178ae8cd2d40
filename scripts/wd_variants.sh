#!/bin/bash
# per-kernel time of the dense wgrad microbenchmark for library variants (scripts/build_variant.sh): usage wd_variants.sh <name>...
# ("base" = the in-tree library).  Output: gpurun_out/wd_variants.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/wd_variants.txt
for v in "$@"; do
  if [ "$v" = base ]; then unset SRCGAN_AMD_LIB; else export SRCGAN_AMD_LIB=$ROOT/srcgan_amd/lib/variants/$v.so; fi
  rm -rf $OUT/prof_wdv
  MB_BLOCKED=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_wdv -- python $ROOT/scripts/microbench_wgrad_dense.py > $OUT/prof_wdv.log 2>&1 || exit 1
  python - "$v" >> $OUT/wd_variants.txt <<PY
import csv, glob, sys, statistics, collections
f = glob.glob("$OUT/prof_wdv/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "wgrad_dense_fast" in r["Kernel_Name"]:
        d[r["Kernel_Name"][:50]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print(f"{sys.argv[1]:10s} {k:50s} median {statistics.median(v):8.1f} us  min {min(v):8.1f}  n={len(v)}")
PY
done
rm -rf $OUT/prof_wdv
cat $OUT/wd_variants.txt
