#!/bin/bash
# Lab sweep of the ping-pong 256 x 256 GEMM tile (csrc/gemm_pp.h; profiles/r06_pp_lab.txt): elimination runs of tools/probe/pp_lab, then the
# library A/B of the family's configurations with bit-identity.   bash tools/pp_lab_sweep.sh [outdir]
#   dbg (timing only): 0 full, 3 MFMAs + barriers only, 6 staging + barriers only;  var 0: LDS-DMA in the load sections, 1: between the MFMAs
O=${1:-gpurun_out/pp_lab}
mkdir -p $O
L=tools/probe/pp_lab
[ -x $L ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value tools/probe/pp_lab.hip -o $L || exit 1
for mx in 0 1; do
  for shape in "12608 3072 768" "12608 768 3072" "4096 4096 4096"; do
    for var in 0 1; do
      for dbg in 0 3 6; do timeout 60 $L $shape $mx $dbg $var; done
    done
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/sweep.txt
for var in 0 1; do
  FERN_PP_VAR=$var timeout 400 python tools/pp_check.py --shapes edge,vit,big --quant 2>&1 | grep -v amdgpu.ids | tee $O/pp_check_var$var.txt
done
