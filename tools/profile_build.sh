#!/bin/bash
# rocprofv3 kernel-trace/stats of one build configuration (default: C3 shape at 2M points), condensed
# into gpurun_out/summary_build_<tag>.txt.  usage: tools/profile_build.sh [config] [tag]
export TMPDIR=/tmp
CFG=${1:-c3:2000000}
TAG=${2:-c3_2m}
O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_build -- python3 tools/run_configs.py $CFG > $O/prof_build_$TAG.json 2> $O/prof_build_$TAG.log || exit 1
python3 tools/prof_summary.py $O/prof_build NOMATCH > $O/summary_build_$TAG.txt 2>&1
rm -rf $O/prof_build
