#!/bin/bash
# more seeds of the three fuzzers (end of round 4)
export TMPDIR=/tmp
o=gpurun_out
( for sd in 401 402 403 404 405 406; do timeout 1200 python tests/fuzz_estep.py --cases 120 --seed $sd 2>&1 | tail -1; done
  for sd in 501 502 503 504; do timeout 1500 python tests/fuzz_update.py --cases 120 --seed $sd 2>&1 | tail -1; done
  for sd in 601 602; do timeout 1500 python tests/fuzz_reference.py --cases 120 --seed $sd 2>&1 | tail -1; done
  for sd in 31 32 33 34; do timeout 900 python tests/fuzz_lifecycle.py --steps 300 --seed $sd 2>&1 | tail -1; done ) | grep -v amdgpu.ids | tee $o/r04_fuzz_wide.txt
