export TMPDIR=/tmp; o=gpurun_out
timeout 2400 python -m pytest tests/test_gpu_update_loop.py tests/test_gpu_merged.py tests/test_gpu_parity.py tests/test_gpu_deferred.py -x -q -m gpu 2>&1 | tail -15 > $o/r06_t2.txt
python3 tools/update_rate.py --configs small,c3 --modes fused --reps 5 2>&1 | grep -v amdgpu.ids > $o/r06_update_rates_c.txt
TRLDA_AUX_DECAY=0 python3 tools/update_rate.py --configs small --modes fused --reps 5 2>&1 | grep -v amdgpu.ids > $o/r06_update_rates_c_nodecay.txt
TRLDA_AUX_DECAY=0 TRLDA_DRAW_AHEAD=0 python3 tools/update_rate.py --configs small --modes fused --reps 5 2>&1 | grep -v amdgpu.ids > $o/r06_update_rates_c_turn.txt
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $o/r06_bench_c.json
cat $o/r06_t2.txt $o/r06_update_rates_c.txt $o/r06_update_rates_c_nodecay.txt $o/r06_update_rates_c_turn.txt
python3 -c "
import json; j=json.loads(open('$o/r06_bench_c.json').read()); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['update_parameters'])"
