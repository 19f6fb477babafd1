"""Per-launch durations of one kernel from a rocprofv3 --kernel-trace CSV, in launch order (which sweep point costs what).
usage: kernel_trace_series.py <dir> <kernel substring> [last N launches]"""
import csv, glob, sys
d, key = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"][:60]))
rows.sort()
if n: rows = rows[-n:]
for i, (t, ms, name) in enumerate(rows):
    print(f"{i:4d} {ms:9.4f} ms  {name}")
print(f"total {sum(r[1] for r in rows):.3f} ms over {len(rows)} launches")
