#!/bin/bash
# dev: bench every prebuilt library variant under julia-newtonraphsonhank_amd/variants/ (box copy is scratch)
P=julia-newtonraphsonhank_amd
cp $P/libhank_hip.so /tmp/lib_base.so
for lib in /tmp/lib_base.so $P/variants/*.so; do
  cp $lib $P/libhank_hip.so
  for N in ${NS:-32 256}; do
    timeout -k 10 200 python bench.py --tangents $N --steps 8 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib) N=$N', round(d['value'],1), 'JVP/s', round(d['ms_per_step'],3), 'ms', d['sweeps_ms'])"
  done
done
cp /tmp/lib_base.so $P/libhank_hip.so
