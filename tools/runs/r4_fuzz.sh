#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out; 
(timeout 900 python tests/fuzz_estep.py --cases 90 --seed 41 2>&1 | tail -4;
 timeout 900 python tests/fuzz_estep.py --cases 90 --seed 42 2>&1 | tail -2;
 timeout 1200 python tests/fuzz_update.py --cases 80 --seed 43 2>&1 | tail -3;
 timeout 1200 python tests/fuzz_update.py --cases 80 --seed 44 2>&1 | tail -2;
 timeout 1200 python tests/fuzz_reference.py --cases 60 --seed 45 2>&1 | tail -2) | grep -v amdgpu.ids | tee $o/r04_fuzz.txt
