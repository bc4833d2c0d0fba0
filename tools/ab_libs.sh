#!/bin/bash
# A/B of library builds / switches inside ONE gpurun call (boxes differ by a few %): every variant is "lib[,VAR=VALUE...]" where
# lib is `default` (the shipped libpann.so) or a name NAME of lib/libpann_NAME.so (make alt ALTNAME=NAME; only those builds read
# the PANN_* switches).  Runs a build config of tools/run_configs.py for each variant, REPS rounds, alternating.
# (replaces the round-1/2 scripts ab_env.sh / ab_envs.sh / ab_leaf.sh / ab_c3.sh, whose switches the shipped library no longer reads:
#  e.g. the leaf-kernel A/B is `CFG=c5:2000000 tools/ab_libs.sh default alt,PANN_LEAF_OLD=1`)
# usage: CFG=c3:2000000 REPS=2 tools/ab_libs.sh default nt alt,PANN_B128_SPLIT=2 nt,PANN_B128_SPLIT=2
CFG=${CFG:-c3:2000000}; REPS=${REPS:-2}
for rep in $(seq $REPS); do
  for spec in "$@"; do
    IFS=, read -r lib envs <<< "$spec"
    unset PANN_LIBRARY
    [ "$lib" != default ] && export PANN_LIBRARY=$GRAFT_REPO_ROOT/parlayann_amd/lib/libpann_$lib.so
    echo "== $spec build $CFG"
    env ${envs//,/ } python3 tools/run_configs.py $CFG 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['build_s'],3), {k: round(v,3) for k,v in j['build_phases_s'].items()}, j.get('recall_at_10', j.get('recall')))"
  done
done
