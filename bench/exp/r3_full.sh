#!/bin/bash
# the whole GPU suite, smoke(), the default bench
set -o pipefail
mkdir -p gpurun_out/r3_full
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread > gpurun_out/r3_full/pytest.log 2>&1 || { tail -40 gpurun_out/r3_full/pytest.log; exit 1; }
tail -3 gpurun_out/r3_full/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_full/smoke.log 2>&1 && tail -2 gpurun_out/r3_full/smoke.log &&
timeout -k 10 300 python bench.py > gpurun_out/r3_full/bench.json 2> gpurun_out/r3_full/bench.err && cat gpurun_out/r3_full/bench.json
