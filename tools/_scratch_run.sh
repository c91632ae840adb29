cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_update_loop.py -x -q -m gpu 2>&1 | tail -2
for args in "--topics 500 --words 100000 --batch 512 --steps 20 --warmup 3" "--topics 500 --words 100000 --batch 4096 --steps 10 --warmup 2" "--topics 200 --words 50000 --batch 12500 --steps 10 --warmup 2" "--topics 150 --words 20000 --batch 4096 --steps 10 --warmup 2" "--topics 300 --words 20000 --batch 4096 --steps 10 --warmup 2"; do
python bench.py --parity-only --no-update-rates $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernels_us'], d['parity']['iteration_counts_equal'], d['parity']['gamma_max_rel_err'])
"
done
