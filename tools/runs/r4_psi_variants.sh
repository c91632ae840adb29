#!/bin/bash
export TMPDIR=/tmp
for v in "" estrin noint both ""; do
  lib=trlda_amd/libtrlda_hip${v:+.$v}.so
  echo "== ${v:-default}"
  for i in 1 2; do TRLDA_LIB=$lib timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --headline-only 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], list(j['roofline']['kernels_us'].values()))"; done
done 2>&1 | tee gpurun_out/r04_psi_variants.txt
