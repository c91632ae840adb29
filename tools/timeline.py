"""Timeline of a rocprofv3 kernel trace (GPU box): per kernel name the mean duration and the mean
idle gap on the device BEFORE it (end of the previous kernel -> its start), and the share of
wall time between the first and the last kernel that no kernel covers.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/update_rate.py ...
    python3 tools/timeline.py DIR [--window name_substring]   # e.g. one update call

`--dump N`: print N consecutive kernels (start offset, duration, gap) from the middle of the trace.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    dump = int(sys.argv[sys.argv.index("--dump") + 1]) if "--dump" in sys.argv else 0
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    if not rows:
        print("no kernel trace under", d)
        return
    dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    busy, prev_end = 0, None
    for s, e, n in rows:
        short = n.split("(")[0].replace("void trlda::", "")[:60]
        dur[short] += e - s
        cnt[short] += 1
        if prev_end is not None and s - prev_end < 200000:      # (longer: the host was elsewhere)
            gap[short] += max(0, s - prev_end)
        busy += e - s
        prev_end = max(prev_end or 0, e)
    print("%-62s %6s %9s %9s" % ("kernel", "calls", "mean us", "gap before"))
    for k in sorted(dur, key=lambda k: -dur[k]):
        print("%-62s %6d %9.2f %9.2f" % (k, cnt[k], dur[k] / cnt[k] / 1e3, gap[k] / cnt[k] / 1e3))
    if dump:
        mid = len(rows) // 2
        t0 = rows[mid][0]
        pe = rows[mid - 1][1]
        for s, e, n in rows[mid:mid + dump]:
            print("%10.2f us  dur %7.2f  gap %6.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - pe) / 1e3,
                                                          n.split("(")[0].replace("void trlda::", "")[:70]))
            pe = e


if __name__ == "__main__":
    main()
