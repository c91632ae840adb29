cd $GRAFT_REPO_ROOT
python bench.py --parity-only --no-update-rates --steps 400 --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pre', d['value'], d['ms_per_step'], list(d['roofline']['kernels_us'].values()), d['parity'])
"
python bench.py --parity-only --no-update-rates --steps 400 --warmup 50 --no-prefetch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('nopre', d['value'], d['ms_per_step'], list(d['roofline']['kernels_us'].values()))
"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
