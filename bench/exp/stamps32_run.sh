set -e
mkdir -p gpurun_out/stamps
export D4PG_HIP_LIB=$PWD/bench/exp/variants/libd4pg_hip_stamps32.so
timeout -k 10 120 python bench/stamps.py ${1:-256} fp32 > gpurun_out/stamps/f32_b${1:-256}.json 2> gpurun_out/stamps/f32_b${1:-256}.err
cat gpurun_out/stamps/f32_b${1:-256}.json
