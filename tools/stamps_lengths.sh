#!/bin/bash
# per-stage cycle shares of the document kernels for batches of equal-length documents (GPU box)
# usage: tools/stamps_lengths.sh "128 144 160 192 193 256 400"
cd "$(dirname "$0")/.." || exit 1
(cd trlda_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DTRLDA_STAMPS -DTRLDA_STAMP_THREAD=${STAMP_THREAD:-0} -o ../libtrlda_hip_stamps.so trlda_hip.hip host_common.cpp host_rng.cpp text_docs.cpp eb_steps.cpp) || exit 1
for n in ${1:-128 144 160 192 193 256 400}; do echo "== n=$n"; STAMPS_LEN=$n python tools/stamps.py; done
