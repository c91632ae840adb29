#!/bin/bash
# tuning aid (GPU box): rows of exp(psi(gamma)) in flight per wave in the statistics kernels.
# Variants are libraries of their own (python -m trlda_amd.build --variant) selected through
# TRLDA_LIB; the package's libtrlda_hip.so is never touched.
i=0
for def in "-DTRLDA_STAT_ROWS=4 -DTRLDA_STAT_ROWS2=4" "-DTRLDA_STAT_ROWS=8 -DTRLDA_STAT_ROWS2=4" "-DTRLDA_STAT_ROWS=8 -DTRLDA_STAT_ROWS2=8" "-DTRLDA_STAT_ROWS=16 -DTRLDA_STAT_ROWS2=8"; do
  i=$((i+1))
  lib=$(python -m trlda_amd.build --variant rows$i $def | tail -1) || exit 1
  export TRLDA_LIB=$lib
  echo "[$def] K=100 B=200:   $(tools/benchline.sh --no-update-rates --repeats 3 --steps 200 --warmup 20 2>&1 | tail -1)"
  echo "[$def] K=200 B=12500: $(tools/benchline.sh --topics 200 --words 50000 --batch 12500 --steps 10 --warmup 2 --no-update-rates --repeats 1 2>&1 | tail -1)"
  echo "[$def] K=500 B=4096:  $(tools/benchline.sh --topics 500 --words 100000 --batch 4096 --steps 10 --warmup 2 --no-update-rates --repeats 1 2>&1 | tail -1)"
  echo "[$def] K=500 B=512:   $(tools/benchline.sh --topics 500 --words 100000 --batch 512 --steps 10 --warmup 2 --no-update-rates --repeats 1 2>&1 | tail -1)"
  echo "[$def] K=100 B=6400:  $(tools/benchline.sh --topics 100 --words 7000 --batch 6400 --steps 20 --warmup 2 --no-update-rates --repeats 1 2>&1 | tail -1)"
  unset TRLDA_LIB
  rm -f "$lib"
done
