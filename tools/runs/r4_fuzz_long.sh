#!/bin/bash
# the long randomised runs of round 4 (E-step, update calls, the reference's own C++), 120 cases per seed
export TMPDIR=/tmp
o=gpurun_out
( for sd in 101 102 103 104; do timeout 1200 python tests/fuzz_estep.py --cases 120 --seed $sd 2>&1 | tail -1; done
  for sd in 201 202 203 204; do timeout 1500 python tests/fuzz_update.py --cases 120 --seed $sd 2>&1 | tail -1; done
  for sd in 301 302; do timeout 1500 python tests/fuzz_reference.py --cases 120 --seed $sd 2>&1 | tail -1; done ) | grep -v amdgpu.ids | tee $o/r04_fuzz_long.txt
timeout 120 tools/probes/event_stream_probe | grep -v "^  after" > $o/r04_event_stream_probe.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $o/r04_gpu_suite.txt
