#!/bin/bash
# several environment variants of one build config inside ONE gpurun call.  usage: tools/ab_envs.sh <config> "VAR=V ..." "VAR=V ..." ...
CFG=$1; shift
for rep in 1 2; do
  for E in "PANN_X=0" "$@"; do
    echo "== [$E] build $CFG"
    env $E python3 tools/run_configs.py $CFG 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['build_s'], j['build_phases_s'], j.get('avg_degree'))"
  done
done
