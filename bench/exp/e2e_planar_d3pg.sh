#!/bin/bash
# The D3PG (scalar critic) configurations of the articulated-body tasks through train.py, two runs at a time.
U=${1:-400000}
bash bench/exp/e2e_planar.sh hopper_d3pg configs/hopper_d3pg.yml $U &
bash bench/exp/e2e_planar.sh bipedal_d3pg configs/bipedal_d3pg.yml $U max_ep_length=1600 &
wait
bash bench/exp/e2e_planar.sh walker2d_d3pg configs/walker2d_d3pg.yml $U &
bash bench/exp/e2e_planar.sh half_cheetah_d3pg configs/half_cheetah_d3pg.yml $U &
wait
