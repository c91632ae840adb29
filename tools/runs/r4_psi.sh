#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_merged.py tests/test_gpu_heavy_tail.py -x -q 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --headline-only 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], list(j['roofline']['kernels_us'].values()))"; done
timeout 300 python tools/update_rate.py --configs small --modes fused 2>&1 | grep -v amdgpu.ids | tail -2
