#!/bin/bash
# Run on the GPU box from the repo root:
#     tools/make_profiles.sh <round-tag> <commit> [target ...]
# Everything the profiles/ directory is made from, as TARGETS (default: all of `core`); each writes
# gpurun_out/<tag>_*, from where what is to be judged is copied into profiles/ (tracked).
#
#   check      smoke, the whole GPU suite, the driver's bench command, the 200-step bench, lifecycle fuzz
#   headline   rocprofv3 --kernel-trace --stats of the default bench command; the un-profiled line
#   pmc        FETCH_SIZE / WRITE_SIZE / SQ_* passes (separate runs, kernel-trace only) + traffic.json
#   k500       the same three for the E-step at K = 500, V = 100 000, B = 512
#   configs    tools/sweep_configs.sh: the other BASELINE.json configurations, with parity
#   updates    tools/update_rate.py, tools/host_rate.py, kernels of the update loops (prof_update.sh)
#   dp         the N > 1 code path on one GPU: forced 1-rank group, --virtual-world 2 / 4 / 8
#   stamps     cycle stamps inside the document kernel (per variant), the deferred launch's timeline,
#              the merged launch's stamps, timelines of an update call and of the bench
#   lengths    tools/length_sweep.py, the reference's test_speed workload, log-normal / uniform bench
#   deferred   the headline with one and two lanes, without deferred statistics / prefetch; helper counts;
#              the two-stream probes
#   probes     graph probe, any-order probe, xcu probe (stand-alone HIP programs under tools/probes)
#   fuzz       the four fuzzers at a few seeds each (long: ~15 min)
#   ingest     round 6: the index builder's host time and phases, the ingestion rates (made / used /
#              streamed), the end-to-end stream call by call
#   lanes      round 6: the hostile set-ups (streams made first, default-priority lanes), round 5's failing
#              case over six process starts
#   core       check headline pmc configs updates dp stamps lengths deferred ingest lanes
tag=${1:-r06}; commit=${2:-unknown}; shift; shift
targets=${@:-core}
export TMPDIR=/tmp
o=gpurun_out
mkdir -p $o
clean() { grep -v "amdgpu.ids"; }
benchline() { python3 bench.py --no-cpu-baseline --no-update-rates "$@" 2>/dev/null | tail -1; }

t_check() {
  ( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
    timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $o/${tag}_bench_driver.json
    python3 - <<PY
import json
j = json.loads(open("$o/${tag}_bench_driver.json").read())
print("driver bench:", j["value"], j["ms_per_step"], "fixed", j.get("value_fixed_work"), "frac", j["roofline"]["frac"],
      "cpu", j["cpu_baseline"]["value"], "tr10", j["update_parameters"]["device_batch_tr10"],
      "tr0", j["update_parameters"]["device_batch_tr0"], "settle", j.get("settle_steps"), j.get("settle_ms"))
PY
    timeout 600 python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > $o/${tag}_bench_200.json
    python3 -c "
import json; j=json.loads(open('$o/${tag}_bench_200.json').read()); print('200 steps:', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernels_us'], j['parity'])"
    for sd in 1 2; do timeout 900 python tests/fuzz_lifecycle.py --steps 300 --seed $sd 2>&1 | tail -1; done
  ) 2>&1 | clean | tee $o/${tag}_final_check.txt
}

t_headline() {
  # the default command: two stream lanes, launches overlap (a launch's duration is about twice the
  # device time per launch); then one launch at a time
  tools/prof_stats.sh ${tag} --steps 200 --warmup 20 > /dev/null
  tools/prof_stats.sh ${tag}_one_lane --lanes 1 --steps 200 --warmup 20 > /dev/null
  cut -c1-160 $o/${tag}_kernel_stats.csv | head -6
  cut -c1-160 $o/${tag}_one_lane_kernel_stats.csv | head -6
  # ... and how the launches of the two lanes lie in time (kernel trace of a short run)
  rm -rf $o/${tag}_ltl; mkdir -p $o/${tag}_ltl
  STEPS=20 rocprofv3 --kernel-trace --output-format csv -d $o/${tag}_ltl -- python3 tools/probes/lanes_dbg.py plain > /dev/null 2>&1
  python3 tools/probes/lanes_timeline.py $o/${tag}_ltl > $o/${tag}_lanes_timeline.txt 2>&1
  rm -rf $o/${tag}_ltl
}

t_pmc() {
  tools/prof_pmc.sh ${tag}_fetch "FETCH_SIZE" --lanes 1 --steps 50 --warmup 5 > $o/${tag}_pmc_fetch.txt
  tools/prof_pmc.sh ${tag}_write "WRITE_SIZE" --lanes 1 --steps 50 --warmup 5 > $o/${tag}_pmc_write.txt
  tools/prof_pmc.sh ${tag}_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" --lanes 1 --steps 50 --warmup 5 > $o/${tag}_pmc_sq.txt
  # the dominant launch: documents + next preamble + previous statistics (estep_merged.h, deferred)
  python3 tools/make_traffic.py $o/${tag}_pmc_fetch.txt $o/${tag}_pmc_write.txt estep_docs_reg_deferred_kernel,estep_docs_tiered_deferred_kernel ${commit} > $o/${tag}_traffic.json
  cat $o/${tag}_pmc_fetch.txt $o/${tag}_pmc_write.txt
}

t_k500() {
  K500="--topics 500 --words 100000 --batch 512 --steps 20 --warmup 3"
  tools/prof_stats.sh ${tag}_k500 $K500 > /dev/null
  tools/prof_pmc.sh ${tag}_k500_fetch "FETCH_SIZE" $K500 > $o/${tag}_k500_pmc_fetch.txt
  tools/prof_pmc.sh ${tag}_k500_write "WRITE_SIZE" $K500 > $o/${tag}_k500_pmc_write.txt
}

t_configs() { bash tools/sweep_configs.sh 2>&1 | clean > $o/${tag}_configs.txt; tail -12 $o/${tag}_configs.txt | cut -c1-200; }

t_updates() {
  python3 tools/update_rate.py --configs small,c3,c5a,c5b,c4 --modes fused,plain 2>&1 | clean > $o/${tag}_update_rates.txt
  ( echo "# TRLDA_AUX_DECAY=0 (the inactive words' decay by the streaming kernel behind the launch)"
    TRLDA_AUX_DECAY=0 python3 tools/update_rate.py --configs small --modes fused 2>&1 | clean | tail -2
    echo "# TRLDA_AUX_DECAY=0 TRLDA_DRAW_AHEAD=0 (every gamma0 drawn in its turn; round 5: 0.427 / 0.087 ms)"
    TRLDA_AUX_DECAY=0 TRLDA_DRAW_AHEAD=0 python3 tools/update_rate.py --configs small --modes fused 2>&1 | clean | tail -2
  ) > $o/${tag}_update_rates_aux_ab.txt
  for cfg in small c5a; do tools/prof_update.sh ${tag}_${cfg}_fused $cfg fused > /dev/null 2>&1; done
  for sl in 1 0; do
    TRLDA_MERGED_SLOTS=$sl python3 tools/update_rate.py --configs small --modes fused 2>&1 | clean | tail -2
  done > $o/${tag}_update_rates_merged_slots.txt
  cat $o/${tag}_update_rates.txt
}

t_dp() {
  for ex in factors sstats; do
    TRLDA_BENCH_FORCE_DIST=1 benchline --steps 200 --warmup 20 --exchange $ex > $o/${tag}_bench_forced_dist_world1_${ex}.json
  done
  for w in 2 4 8; do
    benchline --steps 200 --warmup 20 --virtual-world $w > $o/${tag}_bench_virtual_world${w}.json
    benchline --steps 200 --warmup 20 --virtual-world $w --whole-stats > $o/${tag}_bench_virtual_world${w}_whole_stats.json
  done
  python3 -c "
import json
for w in (2, 4, 8):
    j = json.load(open('$o/${tag}_bench_virtual_world%d.json' % w)); print('virtual world', w, j['ms_per_step'], j['roofline']['kernels_us'])"
}

t_stamps() {
  ( for env in "STAMPS_LEN=100" "STAMPS_LEN=128" "STAMPS_LEN=129" "STAMPS_ONE=129" "STAMPS_ONE=144" "STAMPS_ONE=160"; do
      echo "== $env"; env $env bash tools/stamps.sh 2>&1 | grep -v "amdgpu.ids\|hipcc\|^/"
    done ) > $o/${tag}_stamps_modes.txt 2>&1
  python3 tools/deferred_stamps.py 2>&1 | clean > $o/${tag}_deferred_stamps.txt
  (python3 tools/merged_stamps.py; python3 tools/merged_stamps.py --update) 2>&1 | clean > $o/${tag}_merged_stamps.txt
  for what in update bench; do
    rm -rf $o/${tag}_tl; mkdir -p $o/${tag}_tl
    if [ $what = update ]; then
      rocprofv3 --kernel-trace --output-format csv -d $o/${tag}_tl -- python3 tools/update_rate.py --configs small --modes fused > /dev/null 2>&1
    else
      rocprofv3 --kernel-trace --output-format csv -d $o/${tag}_tl -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-update-rates --headline-only > /dev/null 2>&1
    fi
    python3 tools/timeline.py $o/${tag}_tl --dump 26 > $o/${tag}_timeline_${what}.txt 2>&1
  done
  rm -rf $o/${tag}_tl
  head -12 $o/${tag}_deferred_stamps.txt | cut -c1-250
}

t_lengths() {
  python3 tools/length_sweep.py 2>&1 | clean > $o/${tag}_length_sweep.txt
  python3 tools/length_sweep.py --lanes 1 --lengths 100,128,129,144,145,192,193,400,600 2>&1 | clean > $o/${tag}_length_sweep_one_lane.txt
  python3 tools/length_sweep.py --no-deferred --lengths 100,128,129,144,145,192,193,400 2>&1 | clean > $o/${tag}_length_sweep_no_deferred.txt
  python3 tools/speed_workload.py 2>&1 | clean > $o/${tag}_speed_workload.txt
  for a in "--lengths lognormal" "--uniform"; do
    python3 bench.py --steps 100 --warmup 10 --parity-only --no-update-rates $a 2>/dev/null | tail -1 > $o/${tag}_bench_$(echo $a | tr -d ' -').json
  done
  cat $o/${tag}_length_sweep.txt
  python3 -c "
import json
for n in ('lengthslognormal', 'uniform'):
    j = json.load(open('$o/${tag}_bench_%s.json' % n)); print(n, j['ms_per_step'], j['value'], j['parity'])"
}

t_deferred() {
  ( for a in "" "--lanes 1" "--lanes 1 --no-deferred" "--no-prefetch"; do
      benchline --steps 200 --warmup 20 --headline-only $a | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); print('bench $a:', j['ms_per_step'], j['value'], list(j['roofline']['kernels_us'].values()))"
    done
    for h in 40 56 128; do
      TRLDA_DEFER_HELPERS=$h benchline --steps 200 --warmup 20 --headline-only | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); print('helper workgroups capped at $h:', j['ms_per_step'])"
    done
    TRLDA_LANE_PRIORITY=0 benchline --steps 200 --warmup 20 --headline-only | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); print('lanes on streams of the default priority:', j['ms_per_step'])"
    python3 tools/probes/two_streams.py 2>&1 | grep -v amdgpu.ids
    python3 tools/probes/two_streams_modes.py 2>&1 | grep -v amdgpu.ids ) 2>&1 | tee $o/${tag}_deferred_ab.txt
}

t_probes() {
  python3 tools/graph_probe.py 2>&1 | clean > $o/${tag}_graph_probe.txt
  for p in anyorder_probe xcu_probe; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/$p.hip -o tools/probes/$p 2>/dev/null && tools/probes/$p > $o/${tag}_$p.txt 2>&1
  done
  tail -3 $o/${tag}_graph_probe.txt
}

t_ingest() {
  ( python3 tools/index_rate.py; python3 tools/probes/index_phases.py ) 2>&1 | clean > $o/${tag}_index_rate.txt
  python3 tools/host_rate.py 2>&1 | clean > $o/${tag}_host_rates.txt
  ( export NOGC=1 TRLDA_CALL_TIMES=1
    for a in 4 8; do AHEAD=$a STEPS=6000 python3 tools/probes/e2e_trace.py 2>&1 | clean | grep -v "creates >"; done
    echo "== TRLDA_INDEX_UPLOAD=worker (the workers enqueue the uploads, one at a time)"
    AHEAD=8 TRLDA_INDEX_UPLOAD=worker STEPS=6000 python3 tools/probes/e2e_trace.py 2>&1 | clean | grep -v "creates >"
    echo "== TRLDA_INDEX_THREADS=0 (index and upload inside trlda_batch_create)"
    AHEAD=8 TRLDA_INDEX_THREADS=0 STEPS=3000 python3 tools/probes/e2e_trace.py 2>&1 | clean | grep -v "creates >"
    echo "== resident batches (MODE=nocreate): the probe's documents on the device alone"
    MODE=nocreate STEPS=3000 python3 tools/probes/e2e_trace.py 2>&1 | clean | grep -v "creates >" ) > $o/${tag}_e2e_trace_final.txt
  cat $o/${tag}_index_rate.txt $o/${tag}_host_rates.txt $o/${tag}_e2e_trace_final.txt
}

t_lanes() {
  ( for a in "--hostile 12" "--hostile 0" "--hostile 40" "--hostile 12 --lengths lognormal"; do
      echo "== $a"; python3 tools/probes/lanes_hostile.py $a 2>&1 | clean; done
    echo "== TRLDA_LANE_PRIORITY=0 --hostile 12   (lanes on the default priority, where torch's streams live)"
    TRLDA_LANE_PRIORITY=0 python3 tools/probes/lanes_hostile.py --hostile 12 2>&1 | clean
    echo "== TRLDA_LANE_PRIORITY=0 TRLDA_LANE_VERIFY=0 TRLDA_LANE_CALIBRATE=0 --hostile 12   (round 5's behaviour there)"
    TRLDA_LANE_PRIORITY=0 TRLDA_LANE_VERIFY=0 TRLDA_LANE_CALIBRATE=0 python3 tools/probes/lanes_hostile.py --hostile 12 2>&1 | clean
  ) > $o/${tag}_lanes_hostile.txt
  ( echo "== tools/probes/two_streams.py x 6 process starts (round 5's failing case), the one-lane and two-lane lines"
    for i in 1 2 3 4 5 6; do python3 tools/probes/two_streams.py 2>&1 | clean | tail -3 | head -2 | cut -c1-330; done
    echo "== the same with TRLDA_LANE_VERIFY=0 TRLDA_LANE_CALIBRATE=0 (round 5's library)"
    for i in 1 2; do TRLDA_LANE_VERIFY=0 TRLDA_LANE_CALIBRATE=0 python3 tools/probes/two_streams.py 2>&1 | clean | tail -3 | head -2 | cut -c1-330; done
  ) > $o/${tag}_lanes_two_streams.txt
  cat $o/${tag}_lanes_hostile.txt $o/${tag}_lanes_two_streams.txt
}

t_fuzz() {
  ( for sd in 401 402; do timeout 1200 python tests/fuzz_estep.py --cases 120 --seed $sd 2>&1 | tail -1; done
    for sd in 501 502; do timeout 1500 python tests/fuzz_update.py --cases 120 --seed $sd 2>&1 | tail -1; done
    for sd in 601; do timeout 1500 python tests/fuzz_reference.py --cases 120 --seed $sd 2>&1 | tail -1; done
    for sd in 31 32; do timeout 900 python tests/fuzz_lifecycle.py --steps 300 --seed $sd 2>&1 | tail -1; done ) | clean | tee $o/${tag}_fuzz.txt
}

for t in $targets; do
  if [ $t = core ]; then set -- check headline pmc configs updates dp stamps lengths deferred ingest lanes; else set -- $t; fi
  for u in "$@"; do echo "#### $u"; t_$u; done
done
