#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out
( for sd in 101 102 103 104; do timeout 1200 python tests/fuzz_estep.py --cases 120 --seed $sd 2>&1 | tail -1; done
  for sd in 201 202 203 204; do timeout 1500 python tests/fuzz_update.py --cases 120 --seed $sd 2>&1 | tail -1; done
  for sd in 301 302; do timeout 1500 python tests/fuzz_reference.py --cases 120 --seed $sd 2>&1 | tail -1; done ) | grep -v amdgpu.ids | tee $o/r04_fuzz_long.txt
