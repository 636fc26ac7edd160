#!/bin/bash
# dev: bench a few (workload, N) points
for cfg in "ks_2000x11_T300_N32 16" "ks_2000x11_T300_N32 32" "ks_2000x11_T300_N32 64" "ks_2000x11_T300_N32 128" "ks_2000x11_T300_N32 256" "ks_2000x11_T300_N32 512"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --workload $1 --tangents $2 --steps 5 --warmup 1 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 N=$2', round(d['value'],1), 'JVP/s', round(d['ms_per_step'],3), 'ms', d['sweeps_ms'])"
done
