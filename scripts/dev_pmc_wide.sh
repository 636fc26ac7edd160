# counters of the wide batch (N = 256; HANK_SCHEDULE=launch for the launch path): bash scripts/dev_pmc_wide.sh <tag>
TAG=${1:-r05n256}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/prof_$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARGS="--tangents 256 --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py $ARGS > $O/kt.log 2>&1
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/${TAG}_kernel_stats.csv; rm -rf $O/kt
n=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM"; do
  n=$((n+1))
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/p$n -o pmc -- python3 $R/bench.py $ARGS > $O/p$n.log 2>&1
  python3 $R/scripts/pmc_summary.py $O/p$n | grep -A9 -E "k_fused|k_wide" > $O/${TAG}_pmc_P$n.txt; rm -rf $O/p$n
done
head -4 $O/${TAG}_kernel_stats.csv | cut -c1-160; cat $O/${TAG}_pmc_P*.txt
