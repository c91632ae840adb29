export TMPDIR=/tmp; o=gpurun_out
for env in "A=1"; do
echo "== $env"; env $env STEPS=4000 python3 tools/probes/e2e_trace.py 2>&1 | grep -v amdgpu.ids | tail -3 | head -2 | cut -c1-250; done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['lane_state'], j['lane_calibration'], 'e2e', j['value_end_to_end']['value'], j['value_end_to_end']['one_call']['value'], {k: v['ms_per_call'] for k, v in j['update_parameters'].items() if isinstance(v, dict)})"
python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['lane_state'], 'e2e', j['value_end_to_end']['value'], j['value_end_to_end']['one_call']['value'])"
python3 tools/probes/two_streams.py 2>&1 | grep -v amdgpu.ids | tail -2 | head -1 | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_bench.py tests/test_gpu_lanes.py -x -q -m gpu 2>&1 | tail -3
