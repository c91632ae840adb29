#!/usr/bin/env python
"""Scan the device code for reads the compiler issues ONE AT A TIME: runs of (ds_read | global_load,
s_waitcnt ..cnt(0)) pairs close together -- each pair a full LDS or memory round trip.  Near the
register limit (the tiered document kernels: 246-249 VGPRs) the scheduler does that to reads the
source requests together; round 5 found the change-sum waves of the psi stage that way (927 cycles
against 663: profiles/r05_stamps_waves.txt).  Polling loops show up as short `global` runs and are
what they are.

    python tools/serial_reads_scan.py [min_run=4]      (no GPU needed: hipcc -S of csrc/trlda_hip.hip)
"""
import bisect
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    min_run = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    out = os.path.join(tempfile.gettempdir(), "trlda_device.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics",
                    "-S", "--cuda-device-only", "-w", "trlda_hip.hip", "-o", out],
                   cwd=os.path.join(ROOT, "trlda_amd", "csrc"), check=True)
    lines = open(out).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    ends = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")]
    for s, name in starts:
        e = ends[bisect.bisect_left(ends, s)]
        body = [l.strip().split(";")[0].strip() for l in lines[s:e] if l.strip() and not l.strip().startswith(";")]
        for kind, op, cnt in (("lds", "ds_read", "lgkmcnt(0)"), ("global", "global_load", "vmcnt(0)")):
            hits = [n for n, l in enumerate(body) if l.startswith("s_waitcnt") and cnt in l and
                    any(x.startswith(op) for x in body[max(0, n - 4):n])]
            runs, i = [], 0
            while i < len(hits):
                j = i
                while j + 1 < len(hits) and hits[j + 1] - hits[j] <= 8:
                    j += 1
                if j - i + 1 >= min_run:
                    runs.append((hits[i], j - i + 1))
                i = j + 1
            if runs:
                print("%-6s %-90s %5d instructions, runs (at, reads): %s" % (kind, name[:90], len(body), runs))


if __name__ == "__main__":
    main()
