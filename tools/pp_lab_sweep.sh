#!/bin/bash
# the ping-pong GEMM tile and its software-pipelined form: correctness (library, FERN_PP_VAR) then lab timings.   bash tools/pp_lab_sweep.sh [outdir]
O=${1:-gpurun_out/pp_lab}
mkdir -p $O
FERN_PP_VAR=2 timeout 400 python tools/pp_check.py --shapes edge,vit,big --cfgs 0,2,7 --family bf16 2>&1 | grep -v amdgpu.ids | tee $O/pp_check_var2.txt
FERN_PP_VAR=2 timeout 400 python tools/pp_check.py --shapes edge,vit,big --quant --cfgs 7,11 --family mx8 2>&1 | grep -v amdgpu.ids | tee -a $O/pp_check_var2.txt
L=tools/probe/pp_lab
for mx in 0 1; do
  for shape in "4096 4096 4096" "12608 3072 768" "12608 768 3072"; do
  for var in 1 2; do
    timeout 60 $L $shape $mx 0 $var
  done
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/sweep4.txt
