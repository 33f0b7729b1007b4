#!/bin/bash
# A/B of compile-time kernel variants on ONE GPU box (boxes differ by a few per cent on power-limited kernels):
#   tools/ab_variants.sh <source.hip> "<command>" "<flags of variant A>" "<flags of variant B>" ...
# Each variant: rebuild cone_amd/csrc/<source.hip> with CONE_HIPCC_FLAGS=<flags>, run <command>, print its stdout.
# The default build is restored at the end (also when a variant fails).
src=$1; cmd=$2; shift 2
restore() { touch cone_amd/csrc/$src; CONE_HIPCC_FLAGS="" python3 -m cone_amd.build > /dev/null 2>&1; }
trap restore EXIT
for flags in "$@"; do
    touch cone_amd/csrc/$src
    echo "=== variant: $flags"
    if CONE_HIPCC_FLAGS="$flags" python3 -m cone_amd.build > /tmp/ab_build.log 2>&1; then
        bash -c "$cmd"
    else
        echo "build failed"; tail -5 /tmp/ab_build.log
    fi
done
