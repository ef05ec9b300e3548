#!/bin/bash
# The four planar articulated-body tasks through train.py, two runs at a time (13 processes each on the box's 16 CPUs).
#   bash bench/exp/e2e_planar_all.sh [updates]
U=${1:-400000}
bash bench/exp/e2e_planar.sh hopper_d4pg configs/hopper_d4pg.yml $U v_min=0 v_max=400 &
bash bench/exp/e2e_planar.sh bipedal_d4pg configs/bipedal_d4pg.yml $U max_ep_length=1600 &
wait
bash bench/exp/e2e_planar.sh walker2d_d4pg configs/walker2d_d4pg.yml $U v_min=0 v_max=400 &
bash bench/exp/e2e_planar.sh half_cheetah_d4pg configs/half_cheetah_d4pg.yml $U v_min=-50 v_max=600 &
wait
