#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate
# counter passes (MI355X_MICROARCH.md, HBM section). Run on the GPU box: gpurun -- 'bash scripts/profile_round.sh r01e'
set -o pipefail
TAG=${1:-dev}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extra"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py $ARGS > $O/bench_kt.log 2>&1 || { echo "kernel-trace pass failed"; tail -5 $O/bench_kt.log; exit 1; }
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/${TAG}_kernel_stats.csv
echo "kernel trace done"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $O/$C -o pmc -- python3 $R/bench.py $ARGS > $O/bench_$C.log 2>&1 || { echo "$C pass failed"; tail -5 $O/bench_$C.log; exit 1; }
  python3 $R/scripts/pmc_summary.py $O/$C > $O/${TAG}_pmc_${C}_summary.txt
  echo "$C done"
  rm -rf $O/$C
done
rm -rf $O/kt
head -12 $O/${TAG}_kernel_stats.csv
grep -A3 "k_fused" $O/${TAG}_pmc_*_summary.txt
