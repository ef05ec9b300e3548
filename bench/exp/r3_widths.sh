#!/bin/bash
# compile-time 4x4x1 kernels for the usual hidden widths: tests, then the headline shape at each width
set -o pipefail
O=gpurun_out/r3_widths
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests/test_hip_f32.py tests/test_gpu_rollout.py tests/test_gpu_engine.py -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for h in 64 200 300 512; do
  timeout -k 10 120 python bench.py --no-extras --no-bf16 --hidden $h > $O/bench_lunar256_h$h.json 2> $O/bench_lunar256_h$h.err || exit 1
  timeout -k 10 120 python bench.py --no-extras --no-bf16 --hidden $h --mfma 16x16 > $O/bench_lunar256_h${h}_16x16.json 2> $O/bench_lunar256_h${h}_16x16.err || exit 1
  timeout -k 10 120 python bench.py --no-extras --no-bf16 --hidden $h --layers > $O/bench_lunar256_h${h}_layers.json 2> $O/bench_lunar256_h${h}_layers.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_widths/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['implementation'])
PY
