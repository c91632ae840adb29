#!/bin/bash
# build libtrlda_hip.so with the compiler's per-kernel resource report; print VGPRs / spills / scratch
# of the kernels whose mangled names match $1 (default: the document kernels)
cd "$(dirname "$0")/../trlda_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -munsafe-fp-atomics -Wall \
    -Rpass-analysis=kernel-resource-usage -o ../libtrlda_hip.so trlda_hip.hip host_common.cpp host_rng.cpp text_docs.cpp eb_steps.cpp 2> /tmp/trlda_res.txt || { grep -v "remark:" /tmp/trlda_res.txt | head -40; exit 1; }
grep -v "remark:" /tmp/trlda_res.txt | grep -i "warning\|error" | head
python3 - "${1:-estep_docs}" <<'PY'
import re, sys
txt = open('/tmp/trlda_res.txt').read()
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].split(' [')[0].strip()
    if sys.argv[1] not in name:
        continue
    g = lambda k: (re.search(k + r': (\d+)', b) or [None, '?'])[1]
    print('%-90s VGPR %s spill %s sgpr-spill %s scratch %s' % (name[:90], g('VGPRs'), g('VGPRs Spill'), g('SGPRs Spill'), g(r'ScratchSize \[bytes/lane\]')))
PY
