#!/bin/bash
# usage: tools/prof_pmc.sh <tag> "<counters>" [bench args...]  (GPU box, repo root)
# One PMC pass (its own run, kernel-trace only) over bench.py; prints per-kernel counter means.
tag=$1; shift; ctrs=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-update-rates --headline-only "$@" > $out/bench.log 2>&1
f=$(find $out -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/pmc_${tag}_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "trlda" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("    %-28s mean %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
