#!/bin/bash
# GPU box: smoke, the whole GPU suite, the driver's bench command, the lifecycle fuzz
# usage: tools/runs/r5_check.sh <tag>     -> gpurun_out/<tag>_check.txt
export TMPDIR=/tmp
tag=${1:-r05}
o=gpurun_out
mkdir -p $o
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $o/${tag}_bench_driver.json
  python - <<PY
import json
j = json.loads(open("$o/${tag}_bench_driver.json").read())
print("driver bench:", j["value"], j["ms_per_step"], "fixed", j.get("value_fixed_work"), "frac", j["roofline"]["frac"],
      "cpu", j["cpu_baseline"]["value"], "tr10", j["update_parameters"]["device_batch_tr10"],
      "tr0", j["update_parameters"]["device_batch_tr0"], {k: j[k] for k in j if k.startswith("settle")})
PY
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates 2>/dev/null | tail -1 > $o/${tag}_bench_200.json
  python -c "
import json; j=json.loads(open('$o/${tag}_bench_200.json').read()); print('200 steps:', j['value'], j['ms_per_step'], j['roofline'])"
  for sd in 1 2; do timeout 900 python tests/fuzz_lifecycle.py --steps 300 --seed $sd 2>&1 | tail -2; done
) 2>&1 | grep -v amdgpu.ids | tee $o/${tag}_check.txt
