#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest "tests/test_gpu_update_loop.py::test_fused_update_equals_plain_sequence" -x -q 2>&1 | grep -E "assert|Error|passed|failed" | head -10
rm -rf /tmp/t; mkdir -p /tmp/t
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/t -- python3 tools/update_rate.py --configs small --modes fused > /dev/null 2>&1
python3 tools/timeline.py /tmp/t --dump 14 | head -32
