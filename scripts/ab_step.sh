#!/bin/bash
# Same-box interleaved A/B of the training step between library variants:
#   scripts/ab_step.sh <rounds> <bench args...> -- name=path.so [name=path.so ...]
# e.g. scripts/ab_step.sh 3 --steps 10 --warmup 3 -- old=srcgan_amd/lib/variants/old3x3.so new=srcgan_amd/lib/libsrcgan_amd.so
ROOT=$(cd $(dirname $0)/.. && pwd)
R=$1; shift
ARGS=()
while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    name=${v%%=*}; path=${v#*=}
    ms=$(SRCGAN_AMD_LIB=$ROOT/$path timeout -k 10 300 python $ROOT/bench.py --no-cpu-baseline "${ARGS[@]}" 2>/dev/null | python -c 'import sys,json; print("%.2f" % json.loads(sys.stdin.readline())["ms_per_step"])')
    echo "round $r  $name  $ms ms/step"
  done
done
