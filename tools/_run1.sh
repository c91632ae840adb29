#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
out=gpurun_out/r06_corpus_thread.txt
timeout 900 python -m pytest tests/test_gpu_ingest.py -q -x -m gpu 2>&1 | tail -2 > $out
for rep in 1 2; do for t in 1 0; do
TRLDA_CORPUS_THREAD=$t python bench.py --steps 200 --no-exchange-ab --no-cpu-baseline --no-update-rates 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); e=d['value_end_to_end']; print('maker thread $t:', d['value'], e['value'], e['ms_per_step'], e['one_call']['value'], e['one_call']['ms_per_step'])" >> $out
done; done
