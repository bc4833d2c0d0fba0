#!/bin/bash
# rocprofv3 kernel stats + PMC passes (each in its own process, never with a trace domain) of ANY python tool,
# condensed for the kernels matching <pattern> into gpurun_out/<tag>_{stats,sq,lds,tcc}.txt.
# usage: tools/pmc_cmd.sh <tag> <kernel pattern> <script.py> [args...]
export TMPDIR=/tmp
TAG=$1; PAT=$2; shift 2
O=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_stats -- python3 "$@" > $O/${TAG}_run.log 2>&1 || exit 1
python3 tools/prof_summary.py $O/p_stats $PAT > $O/${TAG}_stats.txt 2>&1; rm -rf $O/p_stats
pass() {
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/p_$name -- python3 "${CMD[@]}" > /dev/null 2> $O/p_$name.log
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  if [ $rc -ne 0 ]; then echo "pass $name failed rc=$rc"; tail -3 $O/p_$name.log; return 0; fi
  python3 tools/prof_summary.py $O/p_$name $PAT > $O/${TAG}_$name.txt 2>&1; rm -rf $O/p_$name
}
CMD=("$@")
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
echo "pmc $TAG done"
