#!/bin/bash
# Run on the GPU box from the repo root:  tools/make_profiles.sh <round-tag> <commit>
# Produces under gpurun_out/ (copy what is to be judged into profiles/):
#   <tag>_kernel_stats.csv       rocprofv3 --kernel-trace --stats of the default bench command
#   <tag>_pmc_fetch.txt / _write.txt / _sq.txt   PMC passes (separate runs, kernel-trace only)
#   <tag>_traffic.json           tools/make_traffic.py from the two passes, stamped with <commit>
#   <tag>_bench.json             the un-profiled default bench line
#   <tag>_k500_*                 the same three for the E-step at K = 500, V = 100 000, B = 512
#   <tag>_configs.txt            tools/sweep_configs.sh: the other BASELINE.json configurations, with parity
#   <tag>_host_rates.txt         tools/host_rate.py: ingestion and PCIe-inclusive entry points
#   <tag>_update_rates.txt       tools/update_rate.py: whole update_parameters calls
#   <tag>_<cfg>_update_kernel_stats.csv   tools/prof_update.sh: kernels of the update loops
#   <tag>_bench_forced_dist_world1_{factors,sstats}.json, <tag>_bench_virtual_world{2,4,8}.json,
#   <tag>_virtual_world8_kernel_stats.csv   the data-parallel step
#   <tag>_stamps_reg.txt         tools/stamps.sh: cycle shares inside the document kernel
#   <tag>_length_sweep.txt, <tag>_speed_workload.txt, <tag>_bench_lengthslognormal.json, <tag>_bench_uniform.json,
#   <tag>_xcu_probe.txt          document lengths
#   <tag>_timeline_*.txt, <tag>_merged_stamps.txt, <tag>_graph_probe.txt, <tag>_anyorder_probe.txt,
#   <tag>_configs_lists_unsplit.txt, <tag>_bench_merged_level{0,2}.json, <tag>_update_rates_unmerged.txt,
#   <tag>_bench_virtual_world*_whole_stats.json    round 4: merged launch, list segments, word-sharded M-step
tag=${1:-r04}; commit=${2:-unknown}
export TMPDIR=/tmp
tools/prof_stats.sh ${tag} --steps 200 --warmup 20 > /dev/null
tools/prof_pmc.sh ${tag}_fetch "FETCH_SIZE" --steps 50 --warmup 5 > gpurun_out/${tag}_pmc_fetch.txt
tools/prof_pmc.sh ${tag}_write "WRITE_SIZE" --steps 50 --warmup 5 > gpurun_out/${tag}_pmc_write.txt
tools/prof_pmc.sh ${tag}_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" --steps 50 --warmup 5 > gpurun_out/${tag}_pmc_sq.txt
python3 tools/make_traffic.py gpurun_out/${tag}_pmc_fetch.txt gpurun_out/${tag}_pmc_write.txt estep_docs_reg_kernel,estep_docs_tiered_kernel ${commit} > gpurun_out/${tag}_traffic.json
K500="--topics 500 --words 100000 --batch 512 --steps 20 --warmup 3"
tools/prof_stats.sh ${tag}_k500 $K500 > /dev/null
tools/prof_pmc.sh ${tag}_k500_fetch "FETCH_SIZE" $K500 > gpurun_out/${tag}_k500_pmc_fetch.txt
tools/prof_pmc.sh ${tag}_k500_write "WRITE_SIZE" $K500 > gpurun_out/${tag}_k500_pmc_write.txt
python3 bench.py --steps 200 --warmup 20 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
bash tools/sweep_configs.sh > gpurun_out/${tag}_configs.txt 2>&1
python3 tools/host_rate.py > gpurun_out/${tag}_host_rates.txt 2>&1
python3 tools/update_rate.py --configs small,c3,c5a,c5b,c4 --modes fused,fused_sep,plain > gpurun_out/${tag}_update_rates.txt 2>&1
python3 tools/update_rate.py --configs small,c5a,c5b,c4 --modes fused --host-draw > gpurun_out/${tag}_update_rates_host_draw.txt 2>&1
# the N > 1 code path on one GPU (1-rank process group), both exchanges; and what ONE rank of 2 / 4 / 8
# executes per step with the factor exchange (--virtual-world: no collective runs)
for ex in factors sstats; do
  TRLDA_BENCH_FORCE_DIST=1 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --exchange $ex 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_forced_dist_world1_${ex}.json
done
TRLDA_BENCH_FORCE_DIST=1 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-update-rates --global-batch 1600 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_forced_dist_world1_b1600.json
for w in 2 4 8; do
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --virtual-world $w 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_virtual_world${w}.json
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --virtual-world $w --whole-stats 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_virtual_world${w}_whole_stats.json
done
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_virtual8_prof -o v8 --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --virtual-world 8 > /dev/null 2>&1
cp $(find gpurun_out/${tag}_virtual8_prof -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_virtual_world8_kernel_stats.csv
bash tools/stamps.sh > gpurun_out/${tag}_stamps_reg.txt 2>&1
# round 4: where a trust-region iteration's time goes (durations back to back, no idle gaps), the
# merged launch from the inside, and the HIP-graph experiment
for mgd in 1 0; do
  rm -rf gpurun_out/${tag}_tl; mkdir -p gpurun_out/${tag}_tl
  TRLDA_MERGED=$mgd rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_tl -- python3 tools/update_rate.py --configs small --modes fused > /dev/null 2>&1
  python3 tools/timeline.py gpurun_out/${tag}_tl --dump 26 > gpurun_out/${tag}_timeline_update_merged${mgd}.txt 2>&1
done
rm -rf gpurun_out/${tag}_tl; mkdir -p gpurun_out/${tag}_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_tl -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --headline-only > /dev/null 2>&1
python3 tools/timeline.py gpurun_out/${tag}_tl --dump 12 > gpurun_out/${tag}_timeline_bench.txt 2>&1
rm -rf gpurun_out/${tag}_tl
(python3 tools/merged_stamps.py; python3 tools/merged_stamps.py --update) 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_merged_stamps.txt
python3 tools/graph_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_graph_probe.txt
TRLDA_SPLIT_LISTS=0 bash tools/sweep_configs.sh > gpurun_out/${tag}_configs_lists_unsplit.txt 2>&1
for mgd in 2 0; do
  TRLDA_MERGED=$mgd python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --headline-only 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_merged_level${mgd}.json
done
TRLDA_MERGED=0 python3 tools/update_rate.py --configs small,c3 --modes fused > gpurun_out/${tag}_update_rates_unmerged.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/anyorder_probe.hip -o tools/probes/anyorder_probe 2>/dev/null && tools/probes/anyorder_probe > gpurun_out/${tag}_anyorder_probe.txt 2>&1
# document lengths: the cliffs between the variants, the reference's own test_speed workload,
# the heavy-tailed and the uniform bench workloads (with their parity legs)
python3 tools/length_sweep.py > gpurun_out/${tag}_length_sweep.txt 2>&1
python3 tools/speed_workload.py > gpurun_out/${tag}_speed_workload.txt 2>&1
for a in "--lengths lognormal" "--uniform"; do
  python3 bench.py --steps 100 --warmup 10 --parity-only --no-update-rates $a 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_$(echo $a | tr -d ' -').json
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/xcu_probe.hip -o tools/probes/xcu_probe 2>/dev/null && tools/probes/xcu_probe > gpurun_out/${tag}_xcu_probe.txt 2>&1
for cfg in small c5a c5b c4; do
  tools/prof_update.sh ${tag}_${cfg}_fused $cfg fused > /dev/null 2>&1
done
tools/prof_update.sh ${tag}_c5a_plain c5a plain > /dev/null 2>&1
python3 tools/allreduce_cost.py 2>/dev/null | grep "K=" > gpurun_out/${tag}_allreduce_world1.txt
cut -c1-160 gpurun_out/${tag}_kernel_stats.csv | head -8
cat gpurun_out/${tag}_pmc_fetch.txt gpurun_out/${tag}_pmc_write.txt
tail -1 gpurun_out/${tag}_bench.json | cut -c1-600
