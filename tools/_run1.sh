export TMPDIR=/tmp; o=gpurun_out
( echo "== tools/probes/two_streams.py x 6 process starts (round 5's failing case: one model on torch's current stream in a process that has made and destroyed streams)"
for i in 1 2 3 4 5 6; do python3 tools/probes/two_streams.py 2>&1 | grep -v amdgpu.ids | tail -3 | head -2 | cut -c1-330; done ) > $o/r06_lanes_two_streams.txt
cat $o/r06_lanes_two_streams.txt
for st in 20 200; do python bench.py --gpus 1 --steps $st --warmup 5 --no-cpu-baseline --no-update-rates --no-end-to-end 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['steps'], j['value'], j['ms_per_step'], j['value_one_lane'], j['lane_state'], j['lane_calibration'], j['settle_steps'])"; done
python3 tools/probes/lanes_hostile.py --hostile 12 2>&1 | grep -v amdgpu.ids | tail -4
TRLDA_LANE_PRIORITY=0 python3 tools/probes/lanes_hostile.py --hostile 12 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_lanes.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-200
