#!/bin/bash
# Build an experimental variant of one kernel file with extra -D flags into srcgan_amd/lib/variants/<name>.so
#   scripts/build_variant.sh <name> <file.hip> -DSG_EXP=1 ...
# Select it at run time with SRCGAN_AMD_LIB=srcgan_amd/lib/variants/<name>.so; NODIAG=1 builds without -DSG_DIAG (production-like: no run-time
# experiment switches), the form to use for A/B timings against the production library (scripts/ab_conv.py).
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
NAME=$1; FILE=$2; shift 2
V=$ROOT/srcgan_amd/lib/variants; mkdir -p $V
DIAG=-DSG_DIAG; if [ -n "$NODIAG" ]; then DIAG=; fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $DIAG "$@" -c $ROOT/srcgan_amd/csrc/$FILE -o $V/$NAME.o
OBJS=""
for o in $ROOT/srcgan_amd/lib/*.o; do
  if [ "$(basename $o .o)" != "$(basename $FILE .hip)" ]; then OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$NAME.so $V/$NAME.o $OBJS
rm $V/$NAME.o
echo $V/$NAME.so
