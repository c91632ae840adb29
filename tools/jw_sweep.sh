#!/bin/bash
# tuning aid (GPU box): build the library with different register-slot counts for the
# single-orientation kernel and time one configuration with each (variant libraries selected
# through TRLDA_LIB; the package's libtrlda_hip.so is never touched)
# usage: tools/jw_sweep.sh "8 9 10" --topics 500 --words 100000 --batch 512 ...
jws="$1"; shift
for jw in $jws; do
  lib=$(python -m trlda_amd.build --variant jw$jw -DTRLDA_WIDE_JW=$jw | tail -1) || exit 1
  echo "JW=$jw: $(TRLDA_LIB=$lib tools/benchline.sh "$@")"
  rm -f "$lib"
done
