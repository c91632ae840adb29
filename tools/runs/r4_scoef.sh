#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out
( for v in "" sc5; do
  lib=trlda_amd/libtrlda_hip${v:+.$v}.so
  echo "== ${v:-default (KS >= 7)}"
  for cfg in "--topics 320 --words 50000 --batch 2048 --steps 10 --warmup 2" "--topics 384 --words 50000 --batch 2048 --steps 10 --warmup 2" "--topics 448 --words 50000 --batch 2048 --steps 10 --warmup 2" "--topics 500 --words 100000 --batch 4096 --steps 10 --warmup 2"; do
    TRLDA_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-update-rates --headline-only $cfg 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('  ', j['ms_per_step'], list(j['roofline']['kernels_us'].values()))"
  done
done ) 2>&1 | tee $o/r04_scoef_ks.txt
