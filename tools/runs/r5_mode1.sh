#!/bin/bash
# GPU box: the 129..144-word register variant -- parity tests that touch it, its stamps, the tiered launch
export TMPDIR=/tmp
tag=${1:-r05}
o=gpurun_out
mkdir -p $o
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_heavy_tail.py tests/test_gpu_merged.py tests/test_gpu_deferred.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -4
  for env in "STAMPS_LEN=129" "STAMPS_ONE=129" "STAMPS_ONE=144"; do
    echo "== $env"; env $env bash tools/stamps.sh 2>&1 | grep -v "amdgpu.ids\|hipcc\|^/"
  done
  rm -rf $o/${tag}_prof; rocprofv3 --kernel-trace --stats -d $o/${tag}_prof -o t --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --headline-only > /dev/null 2>&1
  f=$(find $o/${tag}_prof -name "*kernel_stats.csv" | head -1); cp $f $o/${tag}_kernel_stats.csv; cut -c1-130 $o/${tag}_kernel_stats.csv | head -7
  rm -rf $o/${tag}_prof
) 2>&1 | grep -v amdgpu.ids | tee $o/${tag}_mode1.txt
