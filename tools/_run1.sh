export TMPDIR=/tmp; o=gpurun_out
( echo "== tools/probes/two_streams.py (round 5's failing case: the model on the legacy null stream, streams made and destroyed before)"
  python3 tools/probes/two_streams.py 2>&1 | grep -v amdgpu.ids | tail -3 ) > $o/r06_lanes_null_stream.txt
cat $o/r06_lanes_null_stream.txt
timeout 900 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_bench.py tests/test_gpu_deferred.py -x -q -m gpu 2>&1 | tail -12 | cut -c1-180
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
