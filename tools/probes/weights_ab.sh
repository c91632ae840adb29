# A/B on ONE box: the weights formed by the consuming wave (default) against a thread per word
# through LDS and a barrier (rounds 1-5a; python -m trlda_amd.build --variant waveweights -DTRLDA_WAVE_WEIGHTS; the file below was taken with the roles swapped: "default" = wave weights, "twthread" = a thread per word)
cd $GRAFT_REPO_ROOT
one() { python bench.py "$@" --steps 200 --headline-only --no-update-rates --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])"; }
for rep in 1 2 3; do
  for lib in "" trlda_amd/libtrlda_hip.waveweights.so; do
    echo "lib=${lib:-default} lanes 2: $(TRLDA_LIB=$lib one --lanes 2)  lanes 1: $(TRLDA_LIB=$lib one --lanes 1)  no deferral: $(TRLDA_LIB=$lib one --lanes 1 --no-deferred)"
  done
done
for lib in "" trlda_amd/libtrlda_hip.waveweights.so; do
  echo "lib=${lib:-default}"; TRLDA_LIB=$lib python tools/update_rate.py --configs small,c3 --modes fused 2>&1 | grep -v "tree\|amdgpu"
done
