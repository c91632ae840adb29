#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r4e; rm -rf $o; mkdir -p $o
for mgd in 1 0; do
  export TRLDA_MERGED=$mgd
  timeout 600 python tools/update_rate.py --configs small --modes fused > $o/update_rates_merged$mgd.txt 2>&1
  grep "max_iter_tr" $o/update_rates_merged$mgd.txt
done
export TRLDA_MERGED=1
rm -rf $o/t; mkdir -p $o/t
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $o/t -- python3 tools/update_rate.py --configs small --modes fused > $o/trace_fused.log 2>&1
python3 tools/timeline.py $o/t --dump 24 > $o/timeline_fused.txt; head -40 $o/timeline_fused.txt
rm -rf $o/t
