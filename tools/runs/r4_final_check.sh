#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
timeout 600 python tools/update_rate.py --configs small --modes fused 2>&1 | grep max_iter
TRLDA_MERGED=2 bash tools/runs/r4_quick.sh
bash tools/runs/r4_quick.sh
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read())
print('driver-style', j['value'], j['ms_per_step'], j['repeats'], j['update_parameters']['device_batch_tr10'], j['update_parameters']['device_batch_tr0'])"
