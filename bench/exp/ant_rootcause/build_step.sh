#!/bin/bash
# Builds the variants of ant_step_bisect.hip on the CPU (hipcc cross-compiles) into gpurun_out/ant_rootcause/bin.
D=bench/exp/ant_rootcause; I=d4pg_pytorch_amd/csrc; O=bench/exp/bin/ant_rootcause
mkdir -p $O
g++ -O2 -I$I $D/host_step.cpp -o $O/host_step
b() { tag=$1; shift; hipcc --offload-arch=gfx950 -I$I "$@" $D/ant_step_bisect.hip -o $O/step_$tag -Rpass-analysis=kernel-resource-usage 2> $O/step_$tag.res & }
b O1 -O1
b O2 -O2
b O3 -O3
b O3_restrict -O3 -DWITH_RESTRICT
b O1_hoist -O1 -DHOIST
b O3_hoist -O3 -DHOIST
b O1_hoist_restrict -O1 -DHOIST -DWITH_RESTRICT
b O2_hoist_restrict -O2 -DHOIST -DWITH_RESTRICT
b O3_hoist_restrict -O3 -DHOIST -DWITH_RESTRICT
b O3_hoist_restrict_sgpr_to_mem -O3 -DHOIST -DWITH_RESTRICT -mllvm -amdgpu-spill-sgpr-to-vgpr=false
b O3_hoist_restrict_vgpr_to_mem -O3 -DHOIST -DWITH_RESTRICT -mllvm -amdgpu-spill-vgpr-to-agpr=false
b O3_hoist_restrict_nofma -O3 -DHOIST -DWITH_RESTRICT -ffp-contract=off
b O3_hoist_restrict_notbaa -O3 -DHOIST -DWITH_RESTRICT -fno-strict-aliasing
b O3_hoist_restrict_nounroll -O3 -DHOIST -DWITH_RESTRICT -fno-unroll-loops
b O3_hoist_restrict_dump -O3 -DHOIST -DWITH_RESTRICT -DDUMP_SYSTEM
b O1_dump -O1 -DDUMP_SYSTEM
wait
for f in $O/step_*.res; do echo "$f: $(grep -E 'VGPRs:|AGPRs:|ScratchSize|SGPRs:' $f | sed 's/.*remark: *//;s/ \[-R.*//' | tr '\n' ' ')"; done
