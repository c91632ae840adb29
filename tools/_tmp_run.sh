python tools/length_sweep.py --lengths 145,160,176,192,193 --series all,one > gpurun_out/r03_sweep_tier2_192.txt 2>&1
(cd trlda_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -munsafe-fp-atomics -DTRLDA_TIER2_MAX=144 -o ../libtrlda_hip.so trlda_hip.hip)
python tools/length_sweep.py --lengths 145,160,176,192,193 --series all,one > gpurun_out/r03_sweep_tier2_144.txt 2>&1
grep -v amdgpu gpurun_out/r03_sweep_tier2_192.txt gpurun_out/r03_sweep_tier2_144.txt
