#!/bin/bash
# the run-time knobs of the fp32 headline, again on the current kernels
O=gpurun_out/r3_knobs
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { name=$1; shift; timeout -k 10 100 python bench.py --no-extras --no-bf16 "$@" > $O/$name.json 2> $O/$name.err || return 1; python -c "import json;d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]);print('$name', round(d['ms_per_step']*1000,2))"; }
b base || exit 1
for w in 0 32 96 128; do D4PG_L2_WARMERS=$w b warmers$w || exit 1; done
for w in 8; do D4PG_WGRAD_WAVES=$w b wgrad_waves$w || exit 1; done
D4PG_CRITIC_AHEAD=0 b no_critic_ahead || exit 1
b chunk16 --chunk 16 || exit 1
b chunk64 --chunk 64 || exit 1
b base_again || exit 1
