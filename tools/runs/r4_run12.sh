#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_merged.py tests/test_gpu_update_loop.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4
timeout 900 python tests/fuzz_update.py --cases 40 --seed 77 2>&1 | tail -2
timeout 600 python tools/update_rate.py --configs small --modes fused 2>&1 | grep max_iter
