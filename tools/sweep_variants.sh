#!/bin/bash
# bench.py under each EMP_SWEEP_VARIANT given on the command line (development aid)
for v in "$@"; do
  EMP_SWEEP_VARIANT=$v python bench.py --no-cpu-baseline 2>&1 | tail -1 | V=$v python -c "
import json, os, sys
d = json.loads(sys.stdin.read())
print('variant', os.environ['V'], 'ms/step', d['ms_per_step'], 'value', d['value'], 'frac', d['roofline']['frac'], 'us', d['roofline']['mean_launch_us'], 'diag', d['kernels_ms']['dp_sweep'])"
done
