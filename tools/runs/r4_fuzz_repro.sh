#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/event_stream_probe tools/probes/event_stream_probe.hip && /tmp/event_stream_probe
  timeout 600 python tests/fuzz_estep.py --cases 13 --seed 101 --run 6,12 2>&1 | grep -v amdgpu.ids | tail -3
  timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
  for sd in 101 102; do timeout 1200 python tests/fuzz_estep.py --cases 120 --seed $sd 2>&1 | tail -1; done
) 2>&1 | tee $o/r04_fuzz_repro.txt
