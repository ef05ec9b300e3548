set -e
mkdir -p gpurun_out/stamps
export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_stamps.so
timeout -k 10 120 python bench/stamps.py 256 > gpurun_out/stamps/b256.json 2> gpurun_out/stamps/b256.err
cat gpurun_out/stamps/b256.json
D4PG_L2_WARMERS=0 timeout -k 10 120 python bench/stamps.py 256 > gpurun_out/stamps/b256_nowarm.json 2> gpurun_out/stamps/b256_nowarm.err
cat gpurun_out/stamps/b256_nowarm.json
