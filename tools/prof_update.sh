#!/bin/bash
# usage: tools/prof_update.sh <tag> <configs> <modes>     (GPU box, repo root)
# rocprofv3 kernel-trace + stats of the update loops (tools/update_rate.py); the per-kernel
# summary goes to gpurun_out/<tag>_update_kernel_stats.csv
tag=$1; cfgs=${2:-c5a}; modes=${3:-fused}
export TMPDIR=/tmp
out=gpurun_out/prof_upd_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/update_rate.py --configs $cfgs --modes $modes > $out/run.log 2>&1
f=$(find $out -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${tag}_update_kernel_stats.csv 2>/dev/null
cat $out/run.log | grep -v "^tree"
cut -d, -f1-8 gpurun_out/${tag}_update_kernel_stats.csv | head -24
