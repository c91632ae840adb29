#!/bin/bash
# round 4, GPU call 2: where the time of a trust-region iteration and of a headline step goes:
# kernel-trace timelines (durations AND the idle gaps between kernels), and the statistics + M-step
# + next-preamble kernel with its exp(psi) stream / its group combine taken out (timing only)
export TMPDIR=/tmp
o=gpurun_out/r4b; rm -rf $o; mkdir -p $o
for v in default noexp nogroup; do
  if [ $v != default ]; then export TRLDA_LIB=$PWD/trlda_amd/libtrlda_hip.$v.so; fi
  for mode in fused fused_sep; do
    [ $v != default ] && [ $mode = fused_sep ] && continue
    rm -rf $o/t; mkdir -p $o/t
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d $o/t -- python3 tools/update_rate.py --configs small --modes $mode > $o/trace_${v}_$mode.log 2>&1
    echo "== $v $mode"; grep "max_iter_tr" $o/trace_${v}_$mode.log
    python3 tools/timeline.py $o/t --dump 30 > $o/timeline_${v}_$mode.txt; head -12 $o/timeline_${v}_$mode.txt
  done
  unset TRLDA_LIB
done
rm -rf $o/t; mkdir -p $o/t
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $o/t -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --headline-only > $o/trace_bench.log 2>&1
python3 tools/timeline.py $o/t --dump 12 > $o/timeline_bench.txt; cat $o/timeline_bench.txt
rm -rf $o/t
