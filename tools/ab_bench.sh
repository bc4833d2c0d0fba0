#!/bin/bash
# A/B of two builds of the library on the bench line inside ONE gpurun call (boxes differ by a few %): alternates the default
# libpann.so and $1.  usage: tools/ab_bench.sh <alt.so> [bench args]
ALT=$1; shift
for rep in 1 2 3; do
  for lib in default $ALT; do
    if [ $lib = default ]; then unset PANN_LIBRARY; else export PANN_LIBRARY=$GRAFT_REPO_ROOT/$lib; fi
    echo "== $lib bench $@"
    python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['recall_at_10'], j['roofline']['frac'], j['roofline']['kernel_ms'])"
  done
done
