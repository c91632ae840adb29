#!/bin/bash
# round 4, GPU call 3: the merged launch -- its own tests first (bounded), then the suite, the bench
# with and without it, the trust-region loop with and without it
export TMPDIR=/tmp
o=gpurun_out/r4c; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_merged.py -x -q > $o/pytest_merged.txt 2>&1; echo "merged tests rc=$?"; tail -30 $o/pytest_merged.txt
if grep -q "failed\|error" $o/pytest_merged.txt; then exit 1; fi
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $o/pytest.txt
for mgd in 1 0; do
  export TRLDA_MERGED=$mgd
  timeout 600 python bench.py --steps 200 --warmup 20 --no-update-rates --no-cpu-baseline > $o/bench_200_merged$mgd.json 2> $o/bench_merged$mgd.err
  python3 -c "
import json; j=json.load(open('$o/bench_200_merged$mgd.json'))
print('merged=$mgd', j['value'], j['ms_per_step'], j['repeats'], j['value_no_prefetch'], j['roofline']['kernels_us'])"
  timeout 600 python tools/update_rate.py --configs small --modes fused > $o/update_rates_merged$mgd.txt 2>&1
  grep "max_iter_tr" $o/update_rates_merged$mgd.txt
done
unset TRLDA_MERGED
rm -rf $o/t; mkdir -p $o/t
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $o/t -- python3 tools/update_rate.py --configs small --modes fused > $o/trace_fused.log 2>&1
python3 tools/timeline.py $o/t --dump 24 > $o/timeline_fused.txt; head -40 $o/timeline_fused.txt
rm -rf $o/t
