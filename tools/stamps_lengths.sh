#!/bin/bash
# per-stage cycle shares of the document kernels for batches of equal-length documents (GPU box)
# usage: tools/stamps_lengths.sh "128 144 160 192 193 256 400"
cd "$(dirname "$0")/.." || exit 1
lib=$(python -m trlda_amd.build --variant stamps -DTRLDA_STAMPS -DTRLDA_STAMP_THREAD=${STAMP_THREAD:-0} | tail -1) || exit 1
for n in ${1:-128 144 160 192 193 256 400}; do echo "== n=$n"; TRLDA_LIB=$lib STAMPS_LEN=$n python tools/stamps.py; done
