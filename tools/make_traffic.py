"""profiles/traffic.json from the two PMC summaries of tools/make_profiles.sh:
   python tools/make_traffic.py <fetch summary> <write summary> <doc kernel name> [commit] > profiles/traffic.json
HBM bytes per launch of the dominant kernel, with the gfx950 correction of
MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies 128-B requests at 64 B."""
import json
import sys


def parse(path):
    """per kernel (template variants merged, weighted by their dispatch counts) -> counter means"""
    acc, name = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip().replace("void ", "").split("<")[0]
        elif name:
            parts = line.split()
            n = int(parts[3].strip("()n="))
            tot = acc.setdefault(name, {}).setdefault(parts[0], [0.0, 0])
            tot[0] += float(parts[2]) * n
            tot[1] += n
    out = {k: {c: t[0] / t[1] for c, t in d.items()} for k, d in acc.items()}
    for k, d in acc.items():
        out[k]["_n"] = max(t[1] for t in d.values())
    return out


fetch, write = parse(sys.argv[1]), parse(sys.argv[2])
# the document stage runs under one of several kernel names (the register kernel when no
# document of the batch has more than 128 words, else the tiered one): the mean over all its
# dispatches
names = ["trlda::" + n for n in sys.argv[3].split(",")]
per = {k: {"FETCH_SIZE": fetch[k]["FETCH_SIZE"], "WRITE_SIZE": write.get(k, {}).get("WRITE_SIZE"),
           "dispatches": fetch[k]["_n"]} for k in fetch}
present = [n for n in names if n in per]
tot = sum(per[n]["dispatches"] for n in present)
f = sum(per[n]["FETCH_SIZE"] * per[n]["dispatches"] for n in present) / tot
w = sum(per[n]["WRITE_SIZE"] * per[n]["dispatches"] for n in present) / tot
kernel = present[0] if len(present) == 1 else ",".join(present)
print(json.dumps({
    "measured_at_commit": sys.argv[4] if len(sys.argv) > 4 else None,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, kernel-trace only "
              "(tools/make_profiles.sh); counters are in KB per dispatch, mean over dispatches",
    "workload": "python3 bench.py --no-cpu-baseline --no-update-rates --steps 50 --warmup 5 (K=100, V=7000, 200 documents/step)",
    "kernel": kernel, "kernels": present,
    "fetch_size_kb": f, "write_size_kb": w,
    "correction": "MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, "
                  "i.e. reports half the bytes of a coalesced stream: doubled here; WRITE_SIZE taken as is",
    "hbm_bytes_per_launch": (2 * f + w) * 1024.0,
    "per_kernel_kb": per}, indent=1))
