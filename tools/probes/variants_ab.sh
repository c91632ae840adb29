# same-box A/B of variant builds (python -m trlda_amd.build --variant NAME FLAGS): the headline in its
# three forms, the length sweep's cliffs, update calls
cd $GRAFT_REPO_ROOT
one() { python bench.py "$@" --steps 200 --headline-only --no-update-rates --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])"; }
libs="default $*"
for rep in 1 2; do
  for l in $libs; do
    lib=""; [ $l != default ] && lib=trlda_amd/libtrlda_hip.$l.so
    echo "$l: lanes 2: $(TRLDA_LIB=$lib one --lanes 2)  lanes 1: $(TRLDA_LIB=$lib one --lanes 1)  no deferral: $(TRLDA_LIB=$lib one --lanes 1 --no-deferred)  log-normal lengths: $(TRLDA_LIB=$lib one --lanes 2 --lengths lognormal)"
  done
done
for l in $libs; do
  lib=""; [ $l != default ] && lib=trlda_amd/libtrlda_hip.$l.so
  echo "== $l"; TRLDA_LIB=$lib python tools/update_rate.py --configs small,c3,c5a --modes fused 2>&1 | grep -v "tree\|amdgpu" | cut -c1-110
  TRLDA_LIB=$lib python tools/length_sweep.py --lanes 1 --series all --lengths 100,128,129,144 2>&1 | grep "n="
done
