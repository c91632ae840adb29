#!/bin/bash
# tuning aid (GPU box): compiler scheduling strategies against the default build
cp trlda_amd/libtrlda_hip.so /tmp/libtrlda_hip.orig.so
for fl in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -amdgpu-sched-strategy=iterative-ilp" "-mllvm -amdgpu-sched-strategy=iterative-minreg"; do
  (cd trlda_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -munsafe-fp-atomics $fl -o ../libtrlda_hip.so trlda_hip.hip host_common.cpp host_rng.cpp text_docs.cpp eb_steps.cpp 2>/tmp/flag_err.txt) || { echo "[$fl] build failed: $(tail -2 /tmp/flag_err.txt)"; continue; }
  echo "[$fl] K=100 B=200:   $(tools/benchline.sh --no-update-rates --repeats 3 --steps 200 --warmup 20 2>&1 | tail -1 | cut -c1-230)"
  echo "[$fl] K=200 B=12500: $(tools/benchline.sh --topics 200 --words 50000 --batch 12500 --steps 6 --warmup 2 --no-update-rates --repeats 1 2>&1 | tail -1 | cut -c1-230)"
  echo "[$fl] K=500 B=512:   $(tools/benchline.sh --topics 500 --words 100000 --batch 512 --steps 10 --warmup 2 --no-update-rates --repeats 1 2>&1 | tail -1 | cut -c1-230)"
done
cp /tmp/libtrlda_hip.orig.so trlda_amd/libtrlda_hip.so
