#!/bin/bash
# dev: A/B one environment knob: VAR=name VALS="0 1" NS="32 256" bash scripts/dev_env.sh
for V in $VALS; do for N in ${NS:-32 256}; do
  env $VAR=$V timeout -k 10 200 python bench.py --tangents $N --steps 8 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$V N=$N', round(d['value'],1), 'JVP/s', round(d['ms_per_step'],3), 'ms', d['sweeps_ms'])"
done; done
