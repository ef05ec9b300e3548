#!/bin/bash
# On the GPU box: runs the step variants built by build_step.sh and compares each with the host build.
D=bench/exp/ant_rootcause; B=bench/exp/bin/ant_rootcause; O=gpurun_out/ant_rootcause
mkdir -p $O
$B/host_step $D $O/step_host.bin || exit 1
outs=""
for v in O1 O2 O3 O3_restrict O1_hoist O3_hoist O1_hoist_restrict O2_hoist_restrict O3_hoist_restrict O3_hoist_restrict_sgpr_to_mem O3_hoist_restrict_vgpr_to_mem O3_hoist_restrict_nofma O3_hoist_restrict_notbaa O3_hoist_restrict_nounroll; do
  timeout -k 5 60 $B/step_$v $D $O/step_$v.bin || { echo "step_$v failed"; exit 1; }
  outs="$outs $O/step_$v.bin"
done
python $D/compare_step.py $O/step_host.bin $outs | tee $O/compare_step.txt
for v in O1_dump O3_hoist_restrict_dump; do timeout -k 5 60 $B/step_$v $D $O/step_$v.bin || exit 1; done
python $D/compare_dump.py $O/step_O1_dump.bin $O/step_O3_hoist_restrict_dump.bin | tee $O/compare_dump.txt
