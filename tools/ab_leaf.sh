#!/bin/bash
# A/B of the one-byte leaf kNN kernels inside ONE gpurun call: C5-shaped HCNNG build with the round-1 kernel
# (PANN_LEAF_OLD=1) and with leaf_knn.hip.  usage: tools/ab_leaf.sh [config]
CFG=${1:-c5:2000000}
for rep in 1 2; do
  for mode in new old; do
    if [ $mode = old ]; then E="PANN_LEAF_OLD=1"; else E="PANN_X=1"; fi
    echo "== $mode $CFG"
    env $E python3 tools/run_configs.py $CFG 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['build_s'], j['build_phases_s'], j.get('avg_degree'), j['beam64'])"
  done
done
