#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r4j; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_merged.py tests/test_gpu_dp.py tests/test_gpu_update_loop.py -x -q 2>&1 | tail -4
for cfg in "--topics 100 --words 7000 --batch 1600" "--topics 100 --words 7000 --batch 6400" "--topics 200 --words 50000 --batch 12500 --steps 10 --warmup 2" "--topics 500 --words 100000 --batch 4096 --steps 10 --warmup 2"; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-update-rates --headline-only --repeats 3 $cfg 2>/dev/null | tail -1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read())
print('[$cfg]', j['ms_per_step'], {k[:24]: v for k, v in j['roofline']['kernels_us'].items()})"
done
for w in 8 4 2; do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --virtual-world $w > $o/vw$w.json 2>/dev/null
  python3 -c "
import json; j=json.load(open('$o/vw$w.json'))
print('virtual world $w', j['ms_per_step'], j['roofline']['kernels_us'])"
done
timeout 600 python tools/update_rate.py --configs small,c3 --modes fused 2>&1 | grep max_iter
