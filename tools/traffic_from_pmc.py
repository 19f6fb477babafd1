"""HBM bytes per launch of every kernel of the path from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs (separate
passes, counter unit KiB; gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2, see
/opt/skills/guides/MI355X_MICROARCH.md, HBM section).  usage: traffic_from_pmc.py <outdir> <config> <units per launch> <commit>"""
import csv, glob, json, sys, collections, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from projectultra_amd._lib import source_hash
out = sys.argv[1]
config = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
units = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
commit = sys.argv[4] if len(sys.argv) > 4 else None
# (substring of the kernel name, key in the output) — the key is the kernel CLASS bench.py reports (ultra_hip_kernel_class):
# both decoders (ldpc_totals_kernel for R2/3 .. R5/6, ldpc_decode_kernel otherwise) are "ldpc_decode_kernel" there
ALIASES = {"ldpc_totals_kernel": "ldpc_decode_kernel", "mix_fft2_kernel": "mix_fft_kernel", "track_all_kernel": "track_kernel"}
KEYS = ("ldpc_totals_kernel", "ldpc_decode_kernel", "mix_fft2_kernel", "mix_fft_kernel", "track_all_kernel", "track_pilot_kernel", "track_kernel", "cfo_walk_kernel", "init_state_kernel",
        "count_errors_kernel", "acquire_kernel", "train_kernel")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            name = r["Kernel_Name"]
            for key in KEYS:                       # first match wins: track_pilot_kernel before track_kernel
                if key in name:
                    acc[ALIASES.get(key, key)][c].append(float(r["Counter_Value"]) * 1024.0)
                    break
res = {"config": config, "n_frames": units, "commit": commit, "csrc_sha": source_hash(), "collected": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 1 --warmup 0 --no-build), "
                 "counter unit KiB; gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 128-B requests "
                 "at 64 B, so read bytes = 2 x FETCH_SIZE for coalesced streams; WRITE_SIZE taken as is; mean over launches",
       "kernels": {}}
for k, d in acc.items():
    f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])); w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
    res["kernels"][k] = {"launches": len(d["FETCH_SIZE"]), "FETCH_SIZE_raw_bytes": f, "WRITE_SIZE_bytes": w, "hbm_bytes_per_launch": 2 * f + w}
print(json.dumps(res, indent=1))
