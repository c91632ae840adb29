#!/bin/bash
# one compact line of bench.py output: docs/s, ms/step, per-kernel us
timeout 300 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read())
print(round(j['value']), j['ms_per_step'], j['roofline']['kernels_us'], j['config'].get('mean_iterations_executed'))"
