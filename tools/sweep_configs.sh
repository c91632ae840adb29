for cfg in "--topics 100 --words 7000 --batch 200" "--topics 100 --words 7000 --batch 1600" "--topics 100 --words 7000 --batch 6400" "--topics 200 --words 50000 --batch 12500 --steps 10 --warmup 2" "--topics 200 --words 50000 --batch 12500 --max-iter 100 --steps 5 --warmup 1" "--topics 500 --words 100000 --batch 512 --steps 20 --warmup 3" "--topics 500 --words 100000 --batch 4096 --steps 10 --warmup 2" "--topics 10 --words 1000 --batch 100"; do
  echo "== $cfg"
  timeout 300 python bench.py --steps 50 --warmup 5 --parity-only --no-update-rates $cfg 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read())
print(j['value'], 'docs/s', j['ms_per_step'], 'ms/step', j['roofline']['kernels_us'], 'hbm frac (document launch)', j['roofline']['frac'], 'fp64 frac (document launch)', j['roofline']['fp64']['frac'], 'fp64 TFLOP/s', j['roofline']['fp64']['achieved'], j['roofline']['estep'], j['parity'])"
done
