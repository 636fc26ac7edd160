# dev: the headline bench on several builds of the library (A/B of dev knobs): bash scripts/dev_libs.sh dev/libA.so dev/libB.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out
for L in "$@"; do
  HANK_HIP_LIB=$R/$L timeout -k 10 200 python $R/bench.py --no-extra --no-cpu-baseline > $R/gpurun_out/ab.json 2> $R/gpurun_out/ab.err || { echo "$L: failed"; tail -3 $R/gpurun_out/ab.err; continue; }
  python - "$L" <<PY
import json, sys
d = json.loads(open("$R/gpurun_out/ab.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), d["sweeps_ms"], flush=True)
PY
done
