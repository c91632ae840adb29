#!/usr/bin/env python
"""Where the scalar-register spills of a kernel sit.

The compiler keeps spilled SGPRs in the lanes of a VGPR: `v_writelane_b32 vS, sN, lane` saves,
`v_readlane_b32 sN, vS, lane` restores.  The document kernels use v_readlane themselves (word ids
handed out to the wave), so a restore is recognised by its SOURCE register: a VGPR that is the
target of a v_writelane somewhere in the kernel (the source code has no writelane of its own).

For every kernel whose name matches: the spill VGPRs, the saves / restores in total, and for each
LOOP (a backward branch: label above its branch) the saves / restores inside its body -- the
iteration loop of the fixed point is the loop whose body holds the `v_rcp_f64` of exp(psi).

    python tools/sgpr_spill_scan.py [-v] [name-substring ...]     (no GPU: hipcc -S of csrc/trlda_hip.hip)

Default: the innermost iteration loops only; -v: every loop.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["estep_docs_reg_deferred_kernelILi0", "estep_docs_tiered_deferred_kernelILi2",
           "estep_docs_reg_merged_kernelILi0", "estep_docs_reg_kernelILi0"]


def device_asm(path=None):
    out = path or os.path.join(tempfile.gettempdir(), "trlda_device.s")
    src = os.path.join(ROOT, "trlda_amd", "csrc", "trlda_hip.hip")
    if not os.path.exists(out) or os.path.getmtime(out) < max(
            os.path.getmtime(os.path.join(ROOT, "trlda_amd", "csrc", f))
            for f in os.listdir(os.path.join(ROOT, "trlda_amd", "csrc"))):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
                        "-munsafe-fp-atomics", "-S", "--cuda-device-only", "-w", src, "-o", out],
                       cwd=os.path.dirname(src), check=True)
    return open(out).read().split("\n")


def kernels(lines):
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    ends = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")]
    import bisect
    for s, name in starts:
        yield name, lines[s:ends[bisect.bisect_left(ends, s)]]


def scan(name, body):
    ins = []                                            # (text, label or None)
    labels = {}
    for l in body:
        t = l.split(";")[0].strip()
        if not t:
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if t.startswith("."):
            continue
        ins.append(t)
    spill_regs = set()
    for t in ins:
        m = re.match(r"v_writelane_b32 (v\d+),", t)
        if m:
            spill_regs.add(m.group(1))

    def is_save(t):
        return t.startswith("v_writelane_b32")

    def is_restore(t):
        m = re.match(r"v_readlane_b32 s\d+, (v\d+),", t)
        return bool(m) and m.group(1) in spill_regs

    saves = sum(map(is_save, ins))
    restores = sum(map(is_restore, ins))
    loops = []
    for n, t in enumerate(ins):
        m = re.match(r"s_cbranch_\w+ (\.LBB\w+)|s_branch (\.LBB\w+)", t)
        if not m:
            continue
        lab = m.group(1) or m.group(2)
        if lab in labels and labels[lab] <= n:          # backward: a loop [labels[lab], n]
            lo = labels[lab]
            seg = ins[lo:n + 1]
            loops.append({"at": lo, "len": len(seg), "saves": sum(map(is_save, seg)),
                          "restores": sum(map(is_restore, seg)),
                          "rcp": sum(t2.startswith("v_rcp_f64") for t2 in seg),
                          "barriers": sum(t2.startswith("s_barrier") for t2 in seg),
                          "fp64": sum(bool(re.match(r"v_(fma|mul|add|fmac)_f64", t2)) for t2 in seg)})
    return {"name": name, "instructions": len(ins), "spill_vgprs": sorted(spill_regs), "saves": saves,
            "restores": restores, "loops": loops}


def iteration_loops(r):
    """the loops that ARE a document body's fixed-point iteration (lda.cpp:185-204): an exp(psi)
    (v_rcp_f64) and the stage barriers inside, and no other such loop inside them"""
    cand = [lp for lp in r["loops"] if lp["rcp"] and lp["barriers"] >= 3]
    inner = []
    for lp in cand:
        lo, hi = lp["at"], lp["at"] + lp["len"]
        if not any(o is not lp and lo <= o["at"] and o["at"] + o["len"] <= hi and
                   (o["at"], o["len"]) != (lp["at"], lp["len"]) for o in cand):
            inner.append(lp)
    return inner


def main():
    verbose = "-v" in sys.argv
    pats = [a for a in sys.argv[1:] if a != "-v"] or DEFAULT
    lines = device_asm()
    for name, body in kernels(lines):
        if not any(p in name for p in pats):
            continue
        r = scan(name, body)
        print("%s\n  %d instructions; spill VGPRs %s; %d saves (v_writelane), %d restores (v_readlane from a spill VGPR)"
              % (name, r["instructions"], ",".join(r["spill_vgprs"]) or "-", r["saves"], r["restores"]))
        shown = sorted(r["loops"], key=lambda x: -x["len"]) if verbose else iteration_loops(r)
        for lp in shown:
            if verbose and lp["len"] < 40 and not (lp["saves"] or lp["restores"]):
                continue
            print("    %s at %6d, %5d instructions (%4d fp64 arithmetic, %d barriers, %d v_rcp_f64): "
                  "%3d saves, %3d restores" % ("loop" if verbose else "iteration loop", lp["at"], lp["len"],
                                                 lp["fp64"], lp["barriers"], lp["rcp"], lp["saves"], lp["restores"]))


if __name__ == "__main__":
    main()
