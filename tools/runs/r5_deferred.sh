#!/bin/bash
# GPU box: the deferred-statistics tests, then the headline with and without them, and the kernel trace
export TMPDIR=/tmp
tag=${1:-r05}
o=gpurun_out
mkdir -p $o
( timeout 1200 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_merged.py tests/test_gpu_bench.py -x -q 2>&1 | tail -8
  timeout 900 python -m pytest "tests/test_gpu_update_loop.py" -x -q -k "big_table or announced" 2>&1 | tail -3
  for a in "" "--no-deferred"; do
    timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates $a 2>/dev/null | tail -1 > $o/${tag}_bench_200$a.json
    python - <<PY
import json
j = json.loads(open("$o/${tag}_bench_200$a.json").read())
print("200 steps [$a]:", j["value"], j["ms_per_step"], "noprefetch", j["value_no_prefetch"]["ms_per_step"], "fixed", j["value_fixed_work"]["ms_per_step"],
      j["roofline"]["frac"], j["roofline"]["kernels_us"], j["parity"])
PY
  done
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $o/${tag}_bench_driver.json
  python -c "
import json; j=json.loads(open('$o/${tag}_bench_driver.json').read()); print('driver:', j['value'], j['ms_per_step'], j['roofline']['frac'], j['parity'], j['settle_steps'])"
  rm -rf $o/${tag}_prof; rocprofv3 --kernel-trace --stats -d $o/${tag}_prof -o t --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --headline-only > /dev/null 2>&1
  f=$(find $o/${tag}_prof -name "*kernel_stats.csv" | head -1); cp $f $o/${tag}_kernel_stats.csv; cut -c1-150 $o/${tag}_kernel_stats.csv | head -8
  rm -rf $o/${tag}_prof
  for w in 64 128 512; do
    TRLDA_DEFER_WGS=$w timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --headline-only 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('helpers cap $w:', j['ms_per_step'], j['roofline']['kernels_us'])"
  done
) 2>&1 | grep -v amdgpu.ids | tee $o/${tag}_deferred.txt
