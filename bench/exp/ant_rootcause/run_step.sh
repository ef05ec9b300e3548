#!/bin/bash
# On the GPU box: runs the step variants built by build_step.sh and compares each with the host build.
D=bench/exp/ant_rootcause; B=bench/exp/bin/ant_rootcause; O=gpurun_out/ant_rootcause
mkdir -p $O
$B/host_step $D $O/step_host.bin || exit 1
outs=""
for v in O1 O2 O3 O3_opaque O3_sgpr_to_mem O3_vgpr_to_mem O3_nofma; do
  timeout -k 5 60 $B/step_$v $D $O/step_$v.bin || { echo "step_$v failed"; exit 1; }
  outs="$outs $O/step_$v.bin"
done
python $D/compare_step.py $O/step_host.bin $outs | tee $O/compare_step.txt
