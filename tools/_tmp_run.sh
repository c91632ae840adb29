for w in 8 4; do for it in 20 100; do
echo "== K=200 B=12500 max-iter $it waves $w"
timeout 300 python bench.py --steps 6 --warmup 2 --repeats 3 --parity-only --no-update-rates --topics 200 --words 50000 --batch 12500 --max-iter $it --doc-waves $w 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernels_us'], j['parity'])"
done; done
