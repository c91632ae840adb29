export TMPDIR=/tmp; o=gpurun_out
( for cfg in "--topics 10 --words 1000 --batch 100" "--topics 20 --words 300 --batch 100" "--topics 10 --words 1000 --batch 512 --mean-unique 60" "--topics 10 --words 1000 --batch 1600 --mean-unique 60" "--topics 10 --words 1000 --batch 6400 --mean-unique 60" "--topics 20 --words 7000 --batch 6400 --mean-unique 60" "--topics 32 --words 7000 --batch 6400 --mean-unique 60" "--topics 32 --words 7000 --batch 200"; do
  for sk in 1 0; do echo "== $cfg TRLDA_SMALL_K=$sk"; TRLDA_SMALL_K=$sk timeout 300 python bench.py --steps 30 --warmup 5 --parity-only --no-update-rates --no-end-to-end $cfg 2>/dev/null | tail -1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read()); print(j['value'], 'docs/s', j['ms_per_step'], 'ms/step', j['roofline']['kernels_us'], j['parity']['iteration_counts_equal'])" | cut -c1-250; done; done ) > $o/r06_small_k.txt
cat $o/r06_small_k.txt
