"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE) per kernel template into
profiles/<tag>_pmc_traffic.json.  gfx950 corrections per MI355X_MICROARCH.md (HBM section):
counters are in KiB; FETCH_SIZE reports exactly half the bytes of wide (16 B/lane) coalesced
reads -> doubled; WRITE_SIZE is exact for 16 B/lane streaming stores."""
import csv, glob, json, sys, collections

def load(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out

fetch_dir, write_dir, tag = sys.argv[1], sys.argv[2], sys.argv[3]
fe, wr = load(fetch_dir), load(write_dir)
res = {}
for k in sorted(set(fe) | set(wr)):
    if "svae::" not in k:
        continue
    f = fe.get(k, {}).get("FETCH_SIZE", [])
    w = wr.get(k, {}).get("WRITE_SIZE", [])
    name = k.split("(")[0].replace("void ", "")
    res[name] = {"launches": max(len(f), len(w)),
                 "fetch_bytes_per_launch": (sum(f) / len(f) * 1024 * 2) if f else None,
                 "write_bytes_per_launch": (sum(w) / len(w) * 1024) if w else None}
    if f and w:
        res[name]["hbm_bytes_per_launch"] = res[name]["fetch_bytes_per_launch"] + res[name]["write_bytes_per_launch"]
json.dump({"note": "FETCH_SIZE doubled (gfx950 half-count of 16B/lane reads), KiB->bytes; averaged over launches",
           "kernels": res}, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
for k, v in res.items():
    if v.get("hbm_bytes_per_launch"):
        print(f"{k[:60]:60s} {v['launches']:5d} launches  {v['hbm_bytes_per_launch']/1e6:9.2f} MB/launch")
