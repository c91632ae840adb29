cd trlda_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DTRLDA_STAMPS -DTRLDA_STAMP_THREAD=0 -o ../libtrlda_hip_stamps.so trlda_hip.hip && cd ../..
for n in 128 144 160 192 193 256 400; do echo "== n=$n"; STAMPS_LEN=$n python tools/stamps.py; done
