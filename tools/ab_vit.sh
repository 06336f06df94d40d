#!/bin/bash
# same-box A/B of the ViT forward: the tree in _ab_old/ (a copy of an earlier commit, built) against the working tree.
# Boxes differ by up to 10 %, so only numbers from ONE gpurun call are comparable.   usage: bash tools/ab_vit.sh [model]
M=${1:-ViT-B-32}
for round in 1 2; do
  for tree in _ab_old .; do
    (cd $tree && python tools/vit_bench.py $M 256 --pipe 2>/dev/null | grep "flags=0" | head -2 | sed "s|^|[$tree] |")
  done
done
