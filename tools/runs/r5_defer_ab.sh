#!/bin/bash
# GPU box: the deferred launch's helpers -- tests, timeline of one launch, the headline
export TMPDIR=/tmp
tag=${1:-r05}
o=gpurun_out
mkdir -p $o
( timeout 900 python -m pytest tests/test_gpu_deferred.py -x -q 2>&1 | tail -3
  python tools/deferred_stamps.py 2>&1
  bench() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --headline-only $2 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$1', j['ms_per_step'], list(j['roofline']['kernels_us'].values()))"; }
  bench "default"
  bench "dense preamble" --dense-preamble
  bench "no deferred" --no-deferred
) 2>&1 | grep -v "amdgpu.ids\|hipcc" | tee $o/${tag}_defer_ab.txt
