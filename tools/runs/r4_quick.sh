#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r4q; mkdir -p $o
timeout 300 python bench.py --steps 200 --warmup 20 --no-update-rates --no-cpu-baseline --headline-only --repeats 3 "$@" > $o/b.json 2> $o/b.err
python3 -c "
import json; j=json.load(open('$o/b.json'))
print(j['value'], j['ms_per_step'], j['repeats'], j['roofline']['kernels_us'], j['parity'])"
