#!/bin/bash
# Developer tool: builds the working tree's library with extra compiler flags into tools/_variants/NAME (for tools/gpu_ab*.{py,sh}).
#   tools/build_variant.sh NAME [-DAMC_X_...]...      ENVLINE="export AMC_BLOCKS_PER_CU=10" adds an env file the A/B scripts source
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
B=/tmp/amc_build_$NAME
rm -rf $B; mkdir -p $B/montecarlo_amd $B/include
cp -r $R/montecarlo_amd/csrc $B/montecarlo_amd/csrc
cp $R/include/amc.h $B/include/
rm -f $B/montecarlo_amd/csrc/*.o $B/montecarlo_amd/csrc/*.s $B/montecarlo_amd/csrc/*.gen.h
make -j4 -C $B/montecarlo_amd/csrc EXTRA_HIPFLAGS="$*" OUT=$B/libamc.so 2>&1 | grep -v "warning" | tail -3
V=$R/tools/_variants/$NAME
rm -rf $V; mkdir -p $V/montecarlo_amd
cp $R/montecarlo_amd/*.py $V/montecarlo_amd/
cp $B/libamc.so $V/montecarlo_amd/
[ -n "$ENVLINE" ] && echo "$ENVLINE" > $V/env
rm -rf $B
echo "variant $NAME built"
