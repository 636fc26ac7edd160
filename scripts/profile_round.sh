#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel trace + stats, FETCH_SIZE and WRITE_SIZE in separate counter passes
# (MI355X_MICROARCH.md, HBM section), two SQ passes of <= 8 counters each (the SQ block has 8 slots), for the default
# schedule and — tag suffix "l" — for the per-period launches forced on every entry point (HANK_SCHEDULE=launch).
# Run on the GPU box: gpurun -- 'bash scripts/profile_round.sh r02a'
set -o pipefail
TAG=${1:-dev}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extra"
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
SQ3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU"
run_set() {   # $1 = tag, rest of the environment as set by the caller
  local T=$1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py $ARGS > $O/bench_kt_$T.log 2>&1 || { echo "kernel-trace pass failed"; tail -5 $O/bench_kt_$T.log; return 1; }
  cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/${T}_kernel_stats.csv
  python3 - "$(find $O/kt -name '*kernel_trace.csv' | head -1)" > $O/${T}_kernel_resources.txt <<'PY'
import csv, sys, collections
seen = collections.OrderedDict()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].replace("void hank::", "")
    if k not in seen:
        seen[k] = {c: row.get(c) for c in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
for k, v in seen.items():
    print(k, v)
PY
  grep "^{" $O/bench_kt_$T.log | tail -1 > $O/${T}_bench_line.json
  rm -rf $O/kt
  echo "kernel trace done ($T)"
  local n=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2" "$SQ3"; do
    n=$((n+1))
    local name=$(echo $C | awk '{print $1}'); [ $n -ge 3 ] && name="SQ$((n-2))"
    timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$name -o pmc -- python3 $R/bench.py $ARGS > $O/bench_${name}_$T.log 2>&1 || { echo "$name pass failed"; tail -5 $O/bench_${name}_$T.log; rm -rf $O/pmc_$name; continue; }
    python3 $R/scripts/pmc_summary.py $O/pmc_$name > $O/${T}_pmc_${name}_summary.txt
    echo "$name done ($T)"
    rm -rf $O/pmc_$name
  done
}
run_set $TAG || exit 1
HANK_SCHEDULE=launch run_set ${TAG}l
head -10 $O/${TAG}_kernel_stats.csv
head -6 $O/${TAG}l_kernel_stats.csv
