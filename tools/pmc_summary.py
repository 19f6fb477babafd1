"""Summarise rocprofv3 counter_collection CSVs per kernel (mean per launch)."""
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            k = re.sub(r"^void ", "", k); k = re.sub(r"\(.*", "", k); k = k.replace("ultra_hip::dev::", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[k]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, d in sorted(acc.items()):
    if not any(s in k for s in ("mix_fft", "track", "ldpc_", "init_state", "acquire", "cfo_walk", "count_errors", "chirp", "stimulus", "channel")): continue
    print("==", k, "launches/pass", len(d["_dur_ns"]) // max(1, len(glob.glob(out + "/p*/"))))
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}")
