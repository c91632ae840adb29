#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r4d; rm -rf $o; mkdir -p $o
for dbg in 0 1 2 3 4; do
  export TRLDA_MERGED_DBG=$dbg
  timeout 300 python bench.py --steps 200 --warmup 20 --no-update-rates --no-cpu-baseline --headline-only --repeats 3 > $o/b$dbg.json 2> $o/b$dbg.err
  python3 -c "
import json; j=json.load(open('$o/b$dbg.json'))
print('dbg=$dbg', j['value'], j['ms_per_step'], j['roofline']['kernels_us'])"
done
