cd $GRAFT_REPO_ROOT
one() { python bench.py "$@" --steps 200 --headline-only --no-update-rates --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])"; }
for rep in 1 2 3; do for h in 0 32 40 48; do
  if [ $h = 0 ]; then echo "helpers default (56): $(one --lanes 2)  20 steps: $(python bench.py --steps 20 --warmup 5 --headline-only --no-update-rates --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])")";
  else echo "helpers $h: $(TRLDA_DEFER_HELPERS=$h one --lanes 2)  20 steps: $(TRLDA_DEFER_HELPERS=$h python bench.py --steps 20 --warmup 5 --headline-only --no-update-rates --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])")  log-normal: $(TRLDA_DEFER_HELPERS=$h one --lanes 2 --lengths lognormal)"; fi
done; done
