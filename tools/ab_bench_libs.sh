#!/bin/bash
# A/B of library builds on the default bench workload inside ONE gpurun call: `default` or NAME of lib/libpann_NAME.so; REPS rounds, alternating.
# usage: REPS=2 tools/ab_bench_libs.sh default pv90 pv110
REPS=${REPS:-2}
run() { python3 bench.py --no-hbm-leg --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value']/1e6,3), 'M QPS kernel', round(j['roofline']['kernel_ms'],4), 'ms recall', j['recall_at_10'])"; }
for rep in $(seq $REPS); do
  for lib in "$@"; do
    unset PANN_LIBRARY
    [ "$lib" != default ] && export PANN_LIBRARY=$GRAFT_REPO_ROOT/parlayann_amd/lib/libpann_$lib.so
    echo "== $lib"; run
  done
done
