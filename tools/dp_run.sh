cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dp
export TMPDIR=/tmp
python bench.py --no-cpu-baseline --no-update-rates --steps 400 --warmup 50 > gpurun_out/dp/n1.json 2>gpurun_out/dp/n1.err
TRLDA_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-update-rates --steps 400 --warmup 50 --exchange sstats > gpurun_out/dp/forced_sstats.json 2>gpurun_out/dp/forced_sstats.err
TRLDA_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-update-rates --steps 400 --warmup 50 --exchange factors > gpurun_out/dp/forced_factors.json 2>gpurun_out/dp/forced_factors.err
for w in 2 4 8; do
python bench.py --no-cpu-baseline --no-update-rates --steps 400 --warmup 50 --virtual-world $w > gpurun_out/dp/virtual$w.json 2>gpurun_out/dp/virtual$w.err
done
rocprofv3 --kernel-trace --stats -d gpurun_out/dp/prof8 -o v8 --output-format csv -- python3 bench.py --no-cpu-baseline --no-update-rates --steps 400 --warmup 50 --virtual-world 8 > gpurun_out/dp/prof8.log 2>&1
tail -c 600 gpurun_out/dp/*.err
for f in gpurun_out/dp/*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config'].get('exchange_via'), d['roofline']['kernels_us'])
"; done
find gpurun_out/dp/prof8 -name "*kernel_stats.csv" | head -1 | xargs head -12
