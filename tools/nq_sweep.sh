#!/bin/bash
# throughput of the bench workload vs. queries per step (steady state vs. the 10K-query batch)
for nq in "$@"; do
  python bench.py --no-cpu-baseline --nq $nq --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['nq_per_gpu'], d['value'], d['roofline']['kernel_ms'], d['recall_at_10'], d['roofline']['frac'])"
done
