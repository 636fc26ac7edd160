#!/bin/bash
# dev: lane widths (backward, forward) across batch widths
for LW in ${LWS:-"1 1" "1 2" "2 2"}; do set -- $LW; for N in ${NS:-32 64 256}; do
  HANK_LANE_WIDTH_B=$1 HANK_LANE_WIDTH_F=$2 timeout -k 10 200 python bench.py --tangents $N --steps 8 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LW=$1$2 N=$N', round(d['value'],1), 'JVP/s', round(d['ms_per_step'],3), 'ms', d['sweeps_ms'])"
done; done
