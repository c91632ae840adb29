# A/B on ONE box: exp(psi(x))'s rational part by Horner's rule in u = x (x + 9) (default) against
# Horner's rule in x (python -m trlda_amd.build --variant psix -DTRLDA_PSI_HORNER_X)
cd $GRAFT_REPO_ROOT
one() { python bench.py "$@" --steps 200 --headline-only --no-update-rates --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])"; }
for rep in 1 2 3; do
  for lib in "" trlda_amd/libtrlda_hip.psix.so; do
    echo "lib=${lib:-default} lanes 2: $(TRLDA_LIB=$lib one --lanes 2)  lanes 1: $(TRLDA_LIB=$lib one --lanes 1)  no deferral: $(TRLDA_LIB=$lib one --lanes 1 --no-deferred)"
  done
done
for lib in "" trlda_amd/libtrlda_hip.psix.so; do
  echo "lib=${lib:-default}"; TRLDA_LIB=$lib python tools/update_rate.py --configs small,c3,c5a --modes fused 2>&1 | grep -v "tree\|amdgpu" | cut -c1-110
done
