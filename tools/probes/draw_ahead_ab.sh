cd $GRAFT_REPO_ROOT
for da in 0 1; do
 echo "== TRLDA_DRAW_AHEAD=$da update_rate small"
 TRLDA_DRAW_AHEAD=$da python tools/update_rate.py --configs small,c3 --modes fused 2>&1 | grep -v tree
 TRLDA_DRAW_AHEAD=$da python tools/update_rate.py --configs small --modes fused 2>&1 | grep -v tree
 echo "== TRLDA_DRAW_AHEAD=$da bench"
 TRLDA_DRAW_AHEAD=$da python bench.py --steps 200 | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['update_parameters'])"
 echo "== TRLDA_DRAW_AHEAD=$da GPU_MAX_HW_QUEUES=8 bench"
 GPU_MAX_HW_QUEUES=8 TRLDA_DRAW_AHEAD=$da python bench.py --steps 200 | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['update_parameters'])"
done
