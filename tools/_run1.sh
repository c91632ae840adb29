export TMPDIR=/tmp; o=gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-update-rates 2>/dev/null | tail -1 > $o/r06_bench_e2e.json
python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates 2>/dev/null | tail -1 > $o/r06_bench_e2e_200.json
python bench.py --gpus 1 --steps 1000 --warmup 20 --repeats 3 --no-cpu-baseline --no-update-rates 2>/dev/null | tail -1 > $o/r06_bench_e2e_1000.json
python3 -c "
import json
for f in ('r06_bench_e2e.json','r06_bench_e2e_200.json','r06_bench_e2e_1000.json'):
    j=json.loads(open('$o/'+f).read()); e=j['value_end_to_end']; print(f, j['value'], j['ms_per_step'], 'e2e', e['value'], e['ms_per_step'], e['ms_per_step_min'], 'one lane', j['value_one_lane']['value'])"
