#!/bin/bash
# Matrix-pipe utilisation and shader clock per kernel INSIDE the training step (not a micro-benchmark: back-to-back launches of
# one MFMA-heavy kernel run power-limited at a lower clock than the same kernel does in the step's mix).
#   usage (through gpurun, repo root): bash scripts/pmc_mfma.sh r01      -> gpurun_out/<round>_mfma_pmc.txt
set -e
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY \
    --output-format csv -d $OUT/pmc_mfma_$R -- python $ROOT/bench.py --no-cpu-baseline --no-kernel-profile --steps 2 --warmup 1 > $OUT/pmc_mfma_$R.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_mfma_$R/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:90]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n[k] += 1
with open("$OUT/${R}_mfma_pmc.txt", "w") as w:
    w.write("# rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY\n")
    w.write("#   -- python bench.py --no-cpu-baseline --no-kernel-profile --steps 2 --warmup 1   (round $R; kernels inside the training step)\n")
    w.write("# GRBM_GUI_ACTIVE is summed over the 8 XCDs: /8 / kernel time = shader clock.  SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs) = matrix-pipe busy fraction.\n\n")
    for k in sorted(dur, key=lambda k: -dur[k])[:14]:
        v = agg[k]; us = dur[k] / 1e3
        w.write(f"{k}  launches={n[k]} avg_us={us/n[k]:.1f} total_us={us:.1f}\n")
        for c in sorted(v): w.write(f"     {c:30s} {v[c]:16.0f}   per_us={v[c]/us:12.1f}\n")
        g = v.get("GRBM_GUI_ACTIVE", 0.0)
        if g > 0:
            w.write(f"     => shader clock {g/8/us/1e3:.2f} GHz, matrix pipe busy {100.0*v.get('SQ_VALU_MFMA_BUSY_CYCLES',0.0)/(g/8*1024):.0f} %\n")
PY
rm -rf $OUT/pmc_mfma_$R
grep -E "^[a-zA-Z_]|=>" $OUT/${R}_mfma_pmc.txt
