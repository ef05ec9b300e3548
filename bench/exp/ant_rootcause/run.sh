#!/bin/bash
# Builds ant_bisect.hip at -O1 / -O2 / -O3 (and -O3 without fused multiply-add contraction), runs each on the GPU
# and compares the system each build assembles, and its solution, with the host build.  Run from the repository root.
set -e
D=bench/exp/ant_rootcause
I=d4pg_pytorch_amd/csrc
O=gpurun_out/ant_rootcause
mkdir -p $O
g++ -O2 -I$I $D/host_acc.cpp -o $O/host_acc && $O/host_acc $D $O/acc_host.bin
for L in O1 O2 O3; do
  hipcc -$L --offload-arch=gfx950 -I$I $D/ant_bisect.hip -o $O/ant_bisect_$L
  timeout -k 5 60 $O/ant_bisect_$L $D $O/acc_$L.bin
done
hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I$I $D/ant_bisect.hip -o $O/ant_bisect_O3nofma
timeout -k 5 60 $O/ant_bisect_O3nofma $D $O/acc_O3nofma.bin
python $D/compare.py $O/acc_host.bin $O/acc_O1.bin $O/acc_O2.bin $O/acc_O3.bin $O/acc_O3nofma.bin | tee $O/compare.txt
