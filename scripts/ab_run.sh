#!/bin/bash
# Interleaved A/B of a python script between library variants (separate processes, same box):
#   scripts/ab_run.sh <rounds> <script.py> name=path.so [name=path.so ...]     -- prints every line of every run, prefixed
ROOT=$(cd $(dirname $0)/.. && pwd)
R=$1; S=$2; shift 2
for r in $(seq 1 $R); do
  for v in "$@"; do
    name=${v%%=*}; path=${v#*=}
    SRCGAN_AMD_LIB=$ROOT/$path timeout -k 10 300 python $ROOT/$S 2>/dev/null | sed "s/^/[$r $name] /"
  done
done
