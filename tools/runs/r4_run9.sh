#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r4i; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_update_loop.py -x -q -k "drawn_ahead" 2>&1 | grep -v "^$" | tail -30
for v in default seg512 seg2048; do
  if [ $v != default ]; then export TRLDA_LIB=$PWD/trlda_amd/libtrlda_hip.$v.so; fi
  for cfg in "--topics 100 --words 7000 --batch 1600" "--topics 100 --words 7000 --batch 6400" "--topics 200 --words 50000 --batch 12500 --steps 10 --warmup 2" "--topics 500 --words 100000 --batch 4096 --steps 10 --warmup 2"; do
    timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-update-rates --headline-only --repeats 3 $cfg 2>/dev/null | tail -1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read())
print('$v [$cfg]', j['ms_per_step'], {k[:24]: v for k, v in j['roofline']['kernels_us'].items()})"
  done
  unset TRLDA_LIB
done
timeout 600 python -m pytest tests/test_gpu_merged.py -x -q -k "very_long" 2>&1 | tail -3
