#!/bin/bash
# round 4, GPU call 1: the suite; the bench as the driver runs it and at 200 steps; the trust-region
# loop with the statistics + M-step + next-preamble kernel at <= 64 VGPRs (two 1024-thread
# workgroups per CU) and as 512-thread workgroups at <= 80 (three per CU)
export TMPDIR=/tmp
o=gpurun_out/r4a; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $o/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $o/bench_driver.json 2> $o/bench_driver.err; echo "bench rc=$?"
timeout 600 python bench.py --steps 200 --warmup 20 --no-update-rates > $o/bench_200.json 2> $o/bench_200.err
cut -c1-400 $o/bench_driver.json
for v in default emit512; do
  if [ $v != default ]; then export TRLDA_LIB=$PWD/trlda_amd/libtrlda_hip.$v.so; fi
  timeout 600 python tools/update_rate.py --configs small,c3 --modes fused,fused_sep > $o/update_rates_$v.txt 2>&1
  grep -v "^tree\|amdgpu.ids" $o/update_rates_$v.txt
  rm -rf $o/prof_$v; mkdir -p $o/prof_$v
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$v -- python3 tools/update_rate.py --configs small --modes fused > $o/prof_$v.log 2>&1
  f=$(find $o/prof_$v -name '*kernel_stats.csv' | head -1); cp "$f" $o/small_fused_${v}_kernel_stats.csv; rm -rf $o/prof_$v
  cut -d, -f1-4 $o/small_fused_${v}_kernel_stats.csv | cut -c1-150 | head -8
  unset TRLDA_LIB
done
