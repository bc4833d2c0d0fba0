#!/bin/bash
# Round-2 baseline evidence, one gpurun call:
#  (1) PMC passes of the builder's beam-128 searches (C3 shape, 2M points): kernel stats exist from round 1, counters did not
#  (2) bench.py at the C4-shard size (12.5M x 128 fp16 = 3.2 GB, far beyond the 256 MiB Infinity Cache): kernel stats + PMC
# Each PMC group runs in its own process, never combined with a trace domain.  usage: tools/profile_r02_base.sh [tag]
export TMPDIR=/tmp
TAG=${1:-r02_base}
O=$GRAFT_REPO_ROOT/gpurun_out
CFG=c3:2000000
pmc_build() {   # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/prof_$name -- python3 tools/run_configs.py $CFG > /dev/null 2> $O/prof_$name.log
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then return 1; fi     # timed out / killed: no further GPU step
  if [ $rc -ne 0 ]; then echo "pass $name failed rc=$rc (counter name?)"; tail -3 $O/prof_$name.log; return 0; fi
  python3 tools/prof_summary.py $O/prof_$name beam_search_b128 > $O/${TAG}_b128_$name.txt 2>&1
  rm -rf $O/prof_$name
}
pmc_build fetch FETCH_SIZE || exit 1
pmc_build write WRITE_SIZE || exit 1
pmc_build sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES || exit 1
pmc_build tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum || exit 1
pmc_build lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM || exit 1
echo "b128 pmc done"
BIG="--n 12500000 --no-cpu-baseline"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_big -- python3 bench.py $BIG --steps 20 > $O/${TAG}_big_bench.json 2> $O/${TAG}_big_bench.log || exit 1
python3 tools/prof_summary.py $O/prof_big beam_search > $O/${TAG}_big_stats.txt 2>&1; rm -rf $O/prof_big
for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  set -- $grp; name=$1; shift
  timeout -k 10 600 rocprofv3 --pmc "$@" --output-format csv -d $O/prof_big_$name -- python3 bench.py $BIG --steps 5 > /dev/null 2> $O/prof_big_$name.log || exit 1
  python3 tools/prof_summary.py $O/prof_big_$name beam_search_b64 > $O/${TAG}_big_$name.txt 2>&1; rm -rf $O/prof_big_$name
done
echo "big bench done"
