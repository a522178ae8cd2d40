#!/bin/bash
# One JSON line per BASELINE configuration (paired = the headline, cycle, cas-constlab, x8) -> gpurun_out/bench_<config>_<R>.json
# Usage (GPU box, repo root, through gpurun): scripts/bench_configs.sh r02 [config ...]
R=${1:-r02}; shift
CFGS=${@:-paired cycle cas-constlab x8}
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
mkdir -p $OUT
for c in $CFGS; do
  echo "== $c"
  timeout -k 10 900 python bench.py --config $c --steps 4 --warmup 2 > $OUT/bench_${c}_$R.json 2> $OUT/bench_${c}_$R.err || { echo "FAILED $c"; tail -5 $OUT/bench_${c}_$R.err; }
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${c}_$R.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "step_tflops", "peak_memory_gb", "dtype")}, d["roofline"] and {k: d["roofline"][k] for k in ("kernel", "frac")}, d["cpu_baseline"] and d["cpu_baseline"]["value"])
except Exception as e:
    print("no result:", e)
PY
done
