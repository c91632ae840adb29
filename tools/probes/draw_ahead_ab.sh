# TRLDA_DRAW_AHEAD (the next call's gamma0 on a side stream) with the side stream at the default,
# the high (-1) and the low (1) priority, against drawing in turn; same box
cd $GRAFT_REPO_ROOT
for cfg in "0 0" "1 0" "1 -1" "1 1"; do
  set -- $cfg
  echo "== TRLDA_DRAW_AHEAD=$1 TRLDA_DRAW_PRIORITY=$2"
  TRLDA_DRAW_AHEAD=$1 TRLDA_DRAW_PRIORITY=$2 python tools/update_rate.py --configs small,c3 --modes fused 2>&1 | grep -v "tree\|amdgpu"
done
