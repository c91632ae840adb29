#!/bin/bash
# GPU box: the merged launch's statistics stage in its 16-slot form -- tests, then update calls A/B
export TMPDIR=/tmp
tag=${1:-r05}
o=gpurun_out
mkdir -p $o
( timeout 1800 python -m pytest tests/test_gpu_merged.py tests/test_gpu_update_loop.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4
  for sl in 1 0; do
    echo "== TRLDA_MERGED_SLOTS=$sl"
    TRLDA_MERGED_SLOTS=$sl python tools/update_rate.py --configs small --modes fused 2>&1 | grep -v amdgpu.ids | tail -4
    TRLDA_MERGED_SLOTS=$sl python tools/merged_stamps.py --update 2>&1 | grep -v amdgpu.ids | tail -6
  done
  for sl in 1 0; do
    TRLDA_MERGED_SLOTS=$sl python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('slots $sl: bench update block', j['update_parameters']['device_batch_tr10'], j['update_parameters']['device_batch_tr0'])"
  done
) 2>&1 | tee $o/${tag}_merged_slots.txt
