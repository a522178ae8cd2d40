#!/bin/bash
# Same-box interleaved A/B of whole TREES (e.g. a worktree of an older commit, built in place, against the current tree):
#   scripts/ab_tree.sh <rounds> <treeA> <treeB> <bench args...>
R=$1; A=$2; B=$3; shift 3
for r in $(seq 1 $R); do
  for t in $A $B; do
    ms=$(cd $t && timeout -k 10 600 python bench.py --no-cpu-baseline --no-kernel-profile "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("%.2f ms/step  %.2f /s" % (d["ms_per_step"], d["value"]))')
    echo "round $r  $t  $ms"
  done
done
