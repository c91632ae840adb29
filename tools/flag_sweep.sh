#!/bin/bash
# tuning aid (GPU box): compiler scheduling strategies against the default build.  Every variant is
# a library of its own (python -m trlda_amd.build --variant: the package's flags + the extra ones)
# selected through TRLDA_LIB; the package's libtrlda_hip.so is never touched.
i=0
for fl in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -amdgpu-sched-strategy=iterative-ilp" "-mllvm -amdgpu-sched-strategy=iterative-minreg"; do
  i=$((i+1))
  lib=$(python -m trlda_amd.build --variant sweep$i $fl 2>/tmp/flag_err.txt | tail -1) || { echo "[$fl] build failed: $(tail -2 /tmp/flag_err.txt)"; continue; }
  export TRLDA_LIB=$lib
  echo "[$fl] K=100 B=200:   $(tools/benchline.sh --no-update-rates --repeats 3 --steps 200 --warmup 20 2>&1 | tail -1 | cut -c1-230)"
  echo "[$fl] K=200 B=12500: $(tools/benchline.sh --topics 200 --words 50000 --batch 12500 --steps 6 --warmup 2 --no-update-rates --repeats 1 2>&1 | tail -1 | cut -c1-230)"
  echo "[$fl] K=500 B=512:   $(tools/benchline.sh --topics 500 --words 100000 --batch 512 --steps 10 --warmup 2 --no-update-rates --repeats 1 2>&1 | tail -1 | cut -c1-230)"
  unset TRLDA_LIB
  rm -f "$lib"
done
