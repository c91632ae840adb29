#!/bin/bash
export TMPDIR=/tmp
o=gpurun_out
( for be in 0 1 0 1; do echo "== TRLDA_BIG_EMIT=$be"; TRLDA_BIG_EMIT=$be timeout 600 python tools/update_rate.py --configs c5a,c5b --modes fused 2>&1 | grep "ms/call" | grep "tr=10" | cut -c1-100; done ) | tee $o/r04_big_emit_ab.txt
