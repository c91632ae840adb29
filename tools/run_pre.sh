cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/np
timeout 1500 python -m pytest tests/test_gpu_update_loop.py tests/test_gpu_dp.py -x -q -m gpu 2>&1 | tail -4
python tools/update_rate.py --configs small,c3 --modes fused 2>&1 | grep -v amdgpu.ids
rocprofv3 --kernel-trace --stats -d gpurun_out/np/prof_draw -o p --output-format csv -- python3 tools/update_rate.py --configs small --modes fused > gpurun_out/np/prof_draw.log 2>&1
f=$(find gpurun_out/np/prof_draw -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:12]:
    print("%-70s calls %6s avg %9.1f ns total %10s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["TotalDurationNs"]))
PY
