#!/bin/bash
# tuning aid (GPU box): build the library with different register-slot counts for the
# single-orientation kernel and time one configuration with each
# usage: tools/jw_sweep.sh "8 9 10" --topics 500 --words 100000 --batch 512 ...
jws="$1"; shift
cp trlda_amd/libtrlda_hip.so /tmp/libtrlda_hip.orig.so
for jw in $jws; do
  (cd trlda_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -munsafe-fp-atomics -DTRLDA_WIDE_JW=$jw -o ../libtrlda_hip.so trlda_hip.hip host_common.cpp host_rng.cpp text_docs.cpp eb_steps.cpp) || exit 1
  echo "JW=$jw: $(tools/benchline.sh "$@")"
done
cp /tmp/libtrlda_hip.orig.so trlda_amd/libtrlda_hip.so
