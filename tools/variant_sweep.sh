#!/bin/bash
# C3 (bench.py default workload) under a list of kernelVariant values: one line per variant.   bash tools/variant_sweep.sh 0 16 32 ...
for v in "$@"; do
  python bench.py --variant $v --no-inr --no-k1 --no-scaling-model --no-pipelined --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/vs.json
  python -c "import json; d=json.loads(open('/tmp/vs.json').read()); print('variant $v', d['value'], d['ms_per_step'])"
done
