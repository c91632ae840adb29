#!/bin/bash
# end-of-round check on the GPU box: build entry, smoke, the whole GPU suite, the driver's bench command
export TMPDIR=/tmp
o=gpurun_out
( python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-400
  timeout 600 python bench.py 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('default bench:', j['value'], j['ms_per_step'], j['steps'], j['roofline']['frac'], j['cpu_baseline']['value'], j['update_parameters']['device_batch_tr10'])"
) 2>&1 | grep -v amdgpu.ids | tee $o/r04_final_check.txt
