#!/bin/bash
# Same-box A/B of a kernel source: boxes differ by up to ~5 % on the power-limited kernels, so two gpurun calls cannot
# resolve a 2 % change.  Put the alternative source next to the tree (it must travel with the snapshot, i.e. not under
# gpurun_out/), then on the box:   tools/ab_on_box.sh cone_amd/csrc/ffn_split.hip tools/probe/_ab/variant.hip "<bench command>"
# runs the bench command with the tree's file, swaps the alternative in, rebuilds (hipcc is on the box) and runs it again.
# The committed source is restored on every exit path (also when the B build or run fails).
# (Variants that differ by compile-time flags only: tools/ab_variants.sh.)
set -e
src=$1; alt=$2; cmd=$3
echo "== A: $src as committed"; bash -c "$cmd"
cp "$src" /tmp/ab_keep.hip
trap 'cp /tmp/ab_keep.hip "$src"; python -m cone_amd.build > /dev/null 2>&1' EXIT
cp "$alt" "$src"
python -m cone_amd.build > /dev/null 2>&1
echo "== B: $alt"; bash -c "$cmd"
