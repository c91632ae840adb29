#!/bin/bash
# GPU box: quick A/B -- the listed tests, the 200-step headline, the kernel trace
# usage: tools/runs/r5_quick.sh <tag> "<pytest args>" ["bench args"]
export TMPDIR=/tmp
tag=${1:-r05q}; tests=${2:-tests/test_gpu_deferred.py}; bargs=${3:-}
o=gpurun_out
mkdir -p $o
( timeout 1500 python -m pytest $tests -x -q 2>&1 | tail -6
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --headline-only $bargs 2>/dev/null | tail -1 > $o/${tag}_bench_200.json
  python -c "
import json; j=json.loads(open('$o/${tag}_bench_200.json').read()); print('200 steps:', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernels_us'])"
  rm -rf $o/${tag}_prof; rocprofv3 --kernel-trace --stats -d $o/${tag}_prof -o t --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --headline-only $bargs > /dev/null 2>&1
  f=$(find $o/${tag}_prof -name "*kernel_stats.csv" | head -1); cp $f $o/${tag}_kernel_stats.csv; cut -c1-130 $o/${tag}_kernel_stats.csv | head -7
  rm -rf $o/${tag}_prof
) 2>&1 | grep -v amdgpu.ids | tee $o/${tag}_quick.txt
