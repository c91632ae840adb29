#!/bin/bash
# GPU box: cycle stamps of the document kernel -- 128-word documents (register body <0>), 129 / 144
# (<1>), and the <0> body inside the tiered kernel (one long document in the batch)
export TMPDIR=/tmp
tag=${1:-r05}
o=gpurun_out
mkdir -p $o
( for env in "STAMPS_LEN=100" "STAMPS_LEN=128" "STAMPS_LEN=129" "STAMPS_LEN=144" "STAMPS_ONE=129" "STAMPS_ONE=144" "STAMPS_ONE=160"; do
    echo "== $env"; env $env bash tools/stamps.sh 2>&1 | grep -v "amdgpu.ids\|hipcc\|^/"
  done ) 2>&1 | tee $o/${tag}_stamps_modes.txt
