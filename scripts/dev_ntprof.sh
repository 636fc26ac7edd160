#!/bin/bash
# dev: plain vs nontemporal policy-partials stream in the column-wave persistent sweeps (time + fabric traffic)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/prof_nt; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in plain nt; do
  [ $L = nt ] && export HANK_HIP_LIB=$R/dev/libhank_hip_nt.so || unset HANK_HIP_LIB
  timeout -k 10 200 python3 $R/scripts/dev_wsweep.py col 1 32 64 2>&1 | grep -v amdgpu.ids | sed "s/^/$L /"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $O/p_$L$C -o pmc -- python3 $R/scripts/dev_wsweep.py col 32 > $O/$L$C.log 2>&1
    python3 $R/scripts/pmc_summary.py $O/p_$L$C | grep -A1 "k_xtan" | sed "s/^/$L /"; rm -rf $O/p_$L$C
  done
done
