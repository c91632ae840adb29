import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows))
# last ~70 kernels
t0 = ev[-75][0]
prev_end = None
for s, e, n, q, st in ev[-75:]:
    print("%9.2f %9.2f dur %7.2f  q %s st %s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, n))
