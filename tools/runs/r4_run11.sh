#!/bin/bash
export TMPDIR=/tmp
for tiled in 1 0; do for st in 512 2048 8192; do
  export TRLDA_TILED_TASKS=$tiled TRLDA_SEG_TASKS=$st
  for cfg in "--topics 200 --words 50000 --batch 12500 --steps 10 --warmup 2" "--topics 500 --words 100000 --batch 4096 --steps 10 --warmup 2" "--topics 100 --words 7000 --batch 6400"; do
    timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-update-rates --headline-only --repeats 3 $cfg 2>/dev/null | tail -1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read())
print('tiled=$tiled seg_tasks=$st [$cfg]', j['ms_per_step'], {k[:20]: v for k, v in j['roofline']['kernels_us'].items() if 'sstats' in k})"
  done
done; done
timeout 600 python -m pytest tests/test_gpu_merged.py -x -q -k "very_long" 2>&1 | tail -2
