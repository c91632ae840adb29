#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
out=gpurun_out/r06_k64_skip.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_merged.py tests/test_gpu_deferred.py -q -x -m gpu 2>&1 | tail -2 > $out
b() { python bench.py --steps 200 --headline-only --no-cpu-baseline --no-update-rates "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$TAG $*', d['value'], d['ms_per_step'])" >> $out; }
for rep in 1 2; do
for TAG in new base; do
export TAG
if [ $TAG = base ]; then export TRLDA_LIB=/root/repo/trlda_amd/libtrlda_hip.base.so; else unset TRLDA_LIB; fi
b
b --topics 10 --words 1000 --batch 100 --mean-unique 50
b --topics 20 --words 300 --batch 100 --mean-unique 40
b --topics 50 --words 5000 --batch 200
b --topics 64 --words 7000 --batch 200
done; done
