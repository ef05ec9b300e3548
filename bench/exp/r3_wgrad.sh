#!/bin/bash
# After the software-pipelined weight-gradient loop and the 4-in-flight input loads: numerics, then timings.
O=gpurun_out/r3_wgrad
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_hip_f32.py -m gpu -q --timeout 120 --timeout-method thread -x > $O/tests_f32.log 2>&1; rc=$?; echo "tests exit $rc"; tail -3 $O/tests_f32.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
b() { name=$1; shift; timeout -k 10 150 python bench.py --no-bf16 "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -3 $O/bench_$name.err; return 1; }; python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step']*1000,2),'us', d['implementation'])"; }
b lunar256 --steps 3000 --warmup 300 || exit 1
b lunar256_per --steps 3000 --warmup 300 --per || exit 1
b walker512 --env Walker2d-v2 --batch 512 --steps 2000 --warmup 200 || exit 1
for m in 4x4 16x16; do
b walker1024_$m --env Walker2d-v2 --batch 1024 --steps 2000 --warmup 200 --mfma $m || exit 1
b walker2048_$m --env Walker2d-v2 --batch 2048 --steps 1500 --warmup 200 --mfma $m || exit 1
b walker4096_$m --env Walker2d-v2 --batch 4096 --steps 800 --warmup 100 --mfma $m || exit 1
b ant4096_$m --env Ant-v2 --batch 4096 --steps 800 --warmup 100 --mfma $m || exit 1
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ks() { name=$1; shift; timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/s_$name -o run --output-format csv -- python3 bench.py --no-bf16 --steps 300 --warmup 50 "$@" > $O/${name}_bench.json 2> $O/${name}.err; python tools/rocprof_summary.py stats $O/s_$name $O/kernel_stats_$name.csv; rm -rf $O/s_$name; head -4 $O/kernel_stats_$name.csv | cut -c1-120; }
ks ant4096_16x16 --env Ant-v2 --batch 4096 --mfma 16x16
ks walker2048_4x4 --env Walker2d-v2 --batch 2048 --mfma 4x4
ks lunar256
