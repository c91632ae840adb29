#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out
( for sd in 1 2 3 4 5 6; do timeout 900 python tests/fuzz_lifecycle.py --steps 300 --seed $sd 2>&1 | grep -v amdgpu.ids | tail -2; done ) | tee $o/r04_fuzz_lifecycle.txt
