#!/bin/bash
# A/B inside ONE gpurun call (boxes differ by +-20 %): runs a build config and a bench line alternately without
# and with an environment switch.  usage: tools/ab_env.sh VAR=VALUE <config> [bench args]
SW=$1; CFG=$2; shift; shift
for rep in 1 2; do
  for mode in default "$SW"; do
    echo "== $mode build $CFG"
    if [ "$mode" = default ]; then E=""; else E="$SW"; fi
    env $E python3 tools/run_configs.py $CFG 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['build_s'], j['build_phases_s'])"
    echo "== $mode bench $@"
    env $E python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['recall_at_10'], j['roofline']['frac'], j['roofline']['kernel_ms'])"
  done
done
