#!/bin/bash
# rocprofv3 evidence for a bench.py workload: kernel-trace/stats in one run, each PMC group in its own run (never combined with
# a trace domain), condensed by tools/prof_summary.py into gpurun_out/<tag>_summary_*.txt.
# usage: tools/profile_bench.sh <tag> [bench args...]        (default workload when no args are given)
export TMPDIR=/tmp
TAG=${1:-bench1m}; shift
O=$GRAFT_REPO_ROOT/gpurun_out
run() { # name, rocprof args...
  local name=$1; shift
  timeout -k 10 500 rocprofv3 "$@" --output-format csv -d $O/prof_$name -- python3 bench.py --no-cpu-baseline "${BARGS[@]}" > $O/${TAG}_$name.json 2> $O/${TAG}_$name.log
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name timed out"; exit 1; fi
  if [ $rc -ne 0 ] && [ $rc -ne 3 ]; then echo "pass $name failed rc=$rc"; tail -3 $O/${TAG}_$name.log; return 0; fi
  python3 tools/prof_summary.py $O/prof_$name beam_search > $O/${TAG}_summary_$name.txt 2>&1
  rm -rf $O/prof_$name
}
BARGS=("$@" --steps 20); run stats --kernel-trace --stats
BARGS=("$@" --steps 5)
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
echo "profile $TAG done"
