#!/bin/bash
# round 4, GPU call: the word-sharded M-step -- rank processes on the one GPU, then what ONE rank of
# 2 / 4 / 8 executes per step (--virtual-world) with it and without (--whole-stats)
export TMPDIR=/tmp
o=gpurun_out/r4g; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_dp.py -x -q > $o/pytest_dp.txt 2>&1; echo "dp tests rc=$?"; tail -25 $o/pytest_dp.txt
for w in 8 4 2; do
  for mode in "" "--whole-stats"; do
    timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --virtual-world $w $mode > $o/vw${w}${mode}.json 2> $o/vw.err
    python3 -c "
import json; j=json.load(open('$o/vw${w}${mode}.json'))
print('virtual world $w $mode', j['value'], j['ms_per_step'], j['roofline']['kernels_us'], j['config']['word_sharded_m_step'])"
  done
done
rm -rf $o/t; mkdir -p $o/t
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/t -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --virtual-world 8 > /dev/null 2>&1
cp $(find $o/t -name "*kernel_stats.csv" | head -1) $o/virtual_world8_kernel_stats.csv; rm -rf $o/t
cut -d, -f1-4 $o/virtual_world8_kernel_stats.csv | cut -c1-140 | head -8
