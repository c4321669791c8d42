#!/bin/bash
# bash tools/mx_lab_sweep.sh "<variants>" : every ViT block GEMM of the block-scaled mode through tools/probe/mx_lab, one workgroup per
# tile (grid 0) and persistent (256 or 512 workgroups)
VARS=${1:-"0 1 2 3"}
for v in $VARS; do
  for shp in "12608 3072 768 $v 1 1" "12608 2304 768 $v 0 0" "12608 768 3072 $v 3 0" "12608 768 768 $v 3 0"; do
    for g in 0 256 512; do
      timeout 120 tools/probe/mx_lab $shp $g 2>&1 | grep -v "^bad"
    done
  done
done
