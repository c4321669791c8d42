#!/bin/bash
# elimination experiments of the ping-pong GEMM tile (tools/probe/pp_lab): bash tools/pp_lab_sweep.sh [outdir]
O=${1:-gpurun_out/pp_lab}
mkdir -p $O
L=tools/probe/pp_lab
for mx in 0 1; do
  for shape in "12608 3072 768" "12608 768 3072" "4096 4096 4096"; do
  for var in 0 1; do
  for dbg in 0 3 6 ; do
    timeout 60 $L $shape $mx $dbg $var
  done
  done
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/sweep3.txt
FERN_PP_VAR=1 timeout 300 python tools/pp_check.py --shapes edge,vit,big --quant --cfgs 0,7 --family bf16 2>&1 | grep -v amdgpu.ids | tee $O/pp_check_var1.txt
FERN_PP_VAR=1 timeout 300 python tools/pp_check.py --shapes edge,vit,big --quant --cfgs 7,11 --family mx8 2>&1 | grep -v amdgpu.ids | tee -a $O/pp_check_var1.txt
