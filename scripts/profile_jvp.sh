#!/bin/bash
# kernel trace + FETCH_SIZE / WRITE_SIZE passes of the y-iteration pattern (one hank_primal, eight hank_jvp at N = 32):
# profiles/<tag>_jvp_*.  Run on the GPU box: gpurun -- 'bash scripts/profile_jvp.sh r04b'
set -o pipefail
TAG=${1:-dev}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktj -o kt -- python3 $R/scripts/profile_jvp.py > $O/jvp_kt.log 2>&1 || { tail -5 $O/jvp_kt.log; exit 1; }
cp "$(find $O/ktj -name '*kernel_stats.csv' | head -1)" $O/${TAG}_jvp_kernel_stats.csv; rm -rf $O/ktj
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/pj_$C -o pmc -- python3 $R/scripts/profile_jvp.py > $O/jvp_$C.log 2>&1 || { tail -5 $O/jvp_$C.log; rm -rf $O/pj_$C; continue; }
  python3 $R/scripts/pmc_summary.py $O/pj_$C > $O/${TAG}_jvp_pmc_${C}_summary.txt; rm -rf $O/pj_$C
done
head -8 $O/${TAG}_jvp_kernel_stats.csv | cut -c1-120
grep -A1 "k_xfwd\|k_xtan_back" $O/${TAG}_jvp_pmc_*_summary.txt
