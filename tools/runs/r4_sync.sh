#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r4s; mkdir -p $o
for e in "" "HSA_ENABLE_INTERRUPT=0"; do
  for st in 20 200; do
    env TRLDA_MERGED=0 $e timeout 300 python bench.py --steps $st --warmup 5 --no-update-rates --no-cpu-baseline --headline-only > $o/b.json 2> $o/b.err
    python3 -c "
import json; j=json.load(open('$o/b.json'))
print('[$e] steps=$st', j['value'], j['ms_per_step'], j['repeats'])"
  done
done
