#!/bin/bash
# usage: tools/prof_stats.sh <tag> [bench args...]   (run on the GPU box, from the repo root)
# kernel-trace + stats of bench.py; copies the per-kernel summary to gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-update-rates --headline-only "$@" > $out/bench.log 2>&1
f=$(find $out -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv 2>/dev/null
grep "^{\"metric\"" $out/bench.log | tail -1 > gpurun_out/${tag}_bench.json
cat gpurun_out/${tag}_kernel_stats.csv
