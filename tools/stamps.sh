#!/bin/bash
# diagnostic build + run (GPU box): per-segment cycle shares of the document kernel
cd trlda_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DTRLDA_STAMPS -DTRLDA_STAMP_THREAD=${STAMP_THREAD:-0} -o ../libtrlda_hip_stamps.so trlda_hip.hip host_common.cpp host_rng.cpp text_docs.cpp eb_steps.cpp && cd ../.. && python tools/stamps.py
