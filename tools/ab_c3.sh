#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call (boxes differ by +-20 %): alternates the default
# libpann.so and $1 on a build config and a bench line.  usage: tools/ab_c3.sh <alt.so> [config] [bench args]
ALT=$1; CFG=${2:-c3:2000000}; shift; shift
O=$GRAFT_REPO_ROOT/gpurun_out
for rep in 1 2; do
  for lib in default $ALT; do
    if [ $lib = default ]; then unset PANN_LIBRARY; else export PANN_LIBRARY=$GRAFT_REPO_ROOT/$lib; fi
    echo "== $lib build $CFG"
    python3 tools/run_configs.py $CFG 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['build_s'], j['build_phases_s'])"
    echo "== $lib bench $@"
    python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['frac'], j['roofline']['kernel_ms'])"
  done
done
