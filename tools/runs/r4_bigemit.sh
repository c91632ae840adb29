#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out
timeout 900 python -m pytest tests/test_gpu_update_loop.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
timeout 600 python tests/fuzz_update.py --cases 40 --seed 61 2>&1 | tail -1
timeout 600 python tests/fuzz_reference.py --cases 30 --seed 62 2>&1 | tail -1
( for be in 0 1 0 1; do echo "== TRLDA_BIG_EMIT=$be"; TRLDA_BIG_EMIT=$be timeout 600 python tools/update_rate.py --configs c5a,c5b,c4 --modes fused 2>&1 | grep "ms/call" | cut -c1-100; done ) | tee $o/r04_big_emit_ab.txt
