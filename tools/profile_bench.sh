#!/bin/bash
# rocprofv3 evidence for the bench workload: kernel-trace/stats in one run, each PMC group in its own
# run (never combined with a trace domain), condensed by tools/prof_summary.py into gpurun_out/.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 20 --no-cpu-baseline > $O/prof_stats.json 2> $O/prof_stats.log || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 bench.py --steps 5 --no-cpu-baseline > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 bench.py --steps 5 --no-cpu-baseline > /dev/null 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/prof_sq -- python3 bench.py --steps 5 --no-cpu-baseline > /dev/null 2>&1 || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/prof_tcc -- python3 bench.py --steps 5 --no-cpu-baseline > /dev/null 2>&1 || exit 1
for x in stats fetch write sq tcc; do python3 tools/prof_summary.py $O/prof_$x beam_search > $O/summary_$x.txt 2>&1; done
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq $O/prof_tcc
