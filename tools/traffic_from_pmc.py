"""HBM bytes per launch of the three hot kernels from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs
(separate passes, counter unit KiB; gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2, see
/opt/skills/guides/MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            name = r["Kernel_Name"]
            for key in ("ldpc_decode_kernel", "mix_fft_kernel", "track_kernel"):
                if key in name: acc[key][c].append(float(r["Counter_Value"]) * 1024.0)
res = {"n_frames": 262144,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 1 --warmup 0), "
                 "counter unit KiB; gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 128-B requests "
                 "at 64 B, so read bytes = 2 x FETCH_SIZE for coalesced streams; WRITE_SIZE taken as is; mean over launches",
       "kernels": {}}
for k, d in acc.items():
    f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])); w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
    res["kernels"][k] = {"FETCH_SIZE_raw_bytes": f, "WRITE_SIZE_bytes": w, "hbm_bytes_per_launch": 2 * f + w}
print(json.dumps(res, indent=1))
