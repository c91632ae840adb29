#!/bin/bash
cd trlda_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DTRLDA_STAMPS $1 -o ../libtrlda_hip_stamps.so trlda_hip.hip && cd ../.. && python tools/stamps.py
