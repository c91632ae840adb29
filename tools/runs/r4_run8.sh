#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r4h; rm -rf $o; mkdir -p $o
timeout 1200 python -m pytest tests -m gpu -q -x > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $o/pytest.txt
bash tools/sweep_configs.sh > $o/configs.txt 2>&1; cut -c1-330 $o/configs.txt
TRLDA_SPLIT_LISTS=0 bash tools/sweep_configs.sh > $o/configs_nosplit.txt 2>&1; grep -A1 "12500 --steps 10\|4096\|batch 1600\|batch 6400" $o/configs_nosplit.txt | cut -c1-330
timeout 900 python tools/update_rate.py --configs small,c3,c5a,c5b,c4 --modes fused > $o/update_rates.txt 2>&1; grep max_iter $o/update_rates.txt
