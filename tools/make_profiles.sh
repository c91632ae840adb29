#!/bin/bash
# Run on the GPU box from the repo root:  tools/make_profiles.sh <round-tag>
# Produces under gpurun_out/:
#   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench command
#   <tag>_pmc_fetch.txt / <tag>_pmc_write.txt   FETCH_SIZE / WRITE_SIZE per kernel (separate passes)
#   <tag>_bench.json         the bench line of the profiled run
#   <tag>_configs.txt        tools/sweep_configs.sh: the other BASELINE.json configurations
#   <tag>_host_rates.txt     tools/host_rate.py: PCIe-inclusive entry points
tag=${1:-r01}
export TMPDIR=/tmp
tools/prof_stats.sh ${tag} --steps 200 --warmup 20 > /dev/null
tools/prof_pmc.sh ${tag}_fetch "FETCH_SIZE" --steps 50 --warmup 5 > gpurun_out/${tag}_pmc_fetch.txt
tools/prof_pmc.sh ${tag}_write "WRITE_SIZE" --steps 50 --warmup 5 > gpurun_out/${tag}_pmc_write.txt
tools/prof_pmc.sh ${tag}_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" --steps 50 --warmup 5 > gpurun_out/${tag}_pmc_sq.txt
python3 bench.py --steps 200 --warmup 20 > gpurun_out/${tag}_bench_full.json 2> gpurun_out/${tag}_bench_full.err
# the other BASELINE.json configurations (not bench lines: docs/s + per-kernel us for DESIGN.md)
bash tools/sweep_configs.sh > gpurun_out/${tag}_configs.txt 2>&1
python3 tools/host_rate.py > gpurun_out/${tag}_host_rates.txt 2>&1
cat gpurun_out/${tag}_kernel_stats.csv | cut -c1-160 | head -8
cat gpurun_out/${tag}_pmc_fetch.txt gpurun_out/${tag}_pmc_write.txt
tail -1 gpurun_out/${tag}_bench_full.json
