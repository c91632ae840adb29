#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out
timeout 600 python -m pytest tests/test_gpu_update_loop.py -x -q -k "draw or gamma" 2>&1 | tail -2
( timeout 300 python tools/update_rate.py --configs small,c3 --modes fused 2>&1 | grep -v amdgpu.ids | tail -4 ) | tee $o/r04_draw_fused_ab.txt
timeout 300 bash tools/prof_update.sh r04_draw_fused1 small fused 2>&1 | grep -i "draw\|window\|gamma_sum" | grep -v "^E2026" | cut -d, -f1-4 | cut -c1-40,100-190
