#!/bin/bash
# Converged paths behind profiles/<tag>_solves.jsonl (one JSON line per solve) and the cProfile of the Newton loop.
# Run on the GPU box: gpurun -- 'bash scripts/solves_round.sh r04'
TAG=${1:-dev}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; J=$O/${TAG}_solves.jsonl
: > $J
run() { timeout -k 10 300 python3 "$@" 2>&1 | grep "^{" >> $J; }
run $R/examples/solve_transition.py --n-a 2000 --n-e 11 --cold
run $R/examples/solve_transition.py --n-a 2000 --n-e 11 --cold --inner krylov
run $R/examples/solve_transition.py --n-a 2000 --n-e 11 --cold --jacobian columns
run $R/examples/solve_transition.py --cold
run $R/examples/solve_transition.py --cold --shock 0.8
run $R/examples/solve_transition.py --cold --shock 0.8 --inner krylov
run $R/examples/solve_hank.py
run $R/examples/solve_hank.py --inner krylov
run $R/examples/solve_hank.py --jacobian columns
GRID=2000x11 timeout -k 10 300 python3 $R/scripts/dev_profile_newton.py > $O/${TAG}_newton_ks_2000x11.log 2>&1
MODEL=hank timeout -k 10 300 python3 $R/scripts/dev_profile_newton.py > $O/${TAG}_newton_hank_1000x7.log 2>&1
MODEL=hank INNER=krylov timeout -k 10 300 python3 $R/scripts/dev_profile_newton.py > $O/${TAG}_newton_hank_1000x7_krylov.log 2>&1
wc -l $J
