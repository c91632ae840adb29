#!/bin/bash
# diagnostic build + run (GPU box): per-segment cycle shares of the document kernel
cd "$(dirname "$0")/.." || exit 1
lib=$(python -m trlda_amd.build --variant stamps -DTRLDA_STAMPS -DTRLDA_STAMP_THREAD=${STAMP_THREAD:-0} | tail -1) || exit 1
TRLDA_LIB=$lib python tools/stamps.py
