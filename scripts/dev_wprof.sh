#!/bin/bash
# dev: counters of the slab sweeps at one batch width: bash scripts/dev_wprof.sh TAG N [extra env assignments]
set -o pipefail
TAG=${1:-w}; N=${2:-256}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/scripts/dev_wsweep.py slab $N > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/${TAG}_kernel_stats.csv; rm -rf $O/kt
n=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$((n+1)); name=P$n
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$name -o pmc -- python3 $R/scripts/dev_wsweep.py slab $N > $O/${name}.log 2>&1 || { echo "$name failed"; tail -3 $O/${name}.log; rm -rf $O/pmc_$name; continue; }
  python3 $R/scripts/pmc_summary.py $O/pmc_$name | grep -A12 "k_wtan" > $O/${TAG}_pmc_${name}.txt
  rm -rf $O/pmc_$name
done
head -6 $O/${TAG}_kernel_stats.csv; cat $O/${TAG}_pmc_P*.txt
