#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out/r4f; rm -rf $o; mkdir -p $o
timeout 1200 python -m pytest tests -m gpu -q > $o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 $o/pytest.txt
