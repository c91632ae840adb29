cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/psi2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
python bench.py --parity-only --no-update-rates --steps 400 --warmup 50 > gpurun_out/psi2/n1.json 2>/dev/null
python bench.py --parity-only --no-update-rates --steps 400 --warmup 50 --no-prefetch > gpurun_out/psi2/n1_nopre.json 2>/dev/null
TRLDA_DOC_KERNEL=wide python bench.py --parity-only --no-update-rates --steps 400 --warmup 50 --no-prefetch > gpurun_out/psi2/n1_wide.json 2>/dev/null
for f in gpurun_out/psi2/n1*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernels_us'], d.get('parity'))
"; done
bash tools/stamps.sh 2>&1 | tail -12
