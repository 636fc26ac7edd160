#!/bin/bash
# dev: row groups per wave (backward, forward) across batch widths
for RG in ${RGS:-"1 1" "2 1" "4 2" "2 4"}; do set -- $RG; for N in ${NS:-32 64 256}; do
  HANK_RG_B=$1 HANK_RG_F=$2 timeout -k 10 200 python bench.py --tangents $N --steps 8 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RG=$1$2 N=$N', round(d['value'],1), 'JVP/s', round(d['ms_per_step'],3), 'ms', d['sweeps_ms'])"
done; done
