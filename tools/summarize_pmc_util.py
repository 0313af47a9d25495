"""Per-kernel matrix-core / issue utilisation from a rocprofv3 --pmc pass with
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
-> profiles/<tag>_pmc_mfma.json.

Normalisation (gfx950, 8 XCDs x 32 CUs x 4 SIMDs): GRBM_GUI_ACTIVE is reported summed over the 8 XCD instances, so a
launch lasted GRBM_GUI_ACTIVE / 8 cycles; SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs, so
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024).  Cross-check printed next to it when a FLOP count per launch is
given: mfma_busy should equal (matrix-core FLOP/s) / (dense bf16 peak)."""
import csv, glob, json, sys, collections

d, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "svae::" in k:
            acc[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    if "GRBM_GUI_ACTIVE" not in m or "SQ_VALU_MFMA_BUSY_CYCLES" not in m:
        continue
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    wave = max(m.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    res[k] = {"launches": len(cs["GRBM_GUI_ACTIVE"]), "cycles_per_launch": round(cyc),
              "mfma_busy": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4),
              "wave_cycles_waiting_on_waitcnt": round(m.get("SQ_WAIT_ANY", 0.0) / wave, 4),
              "wave_cycles_waiting_to_issue": round(m.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4),
              "wave_cycles_issuing": round(m.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 4),
              "lds_bank_conflict_cycles_per_launch": round(m.get("SQ_LDS_BANK_CONFLICT", 0.0))}
json.dump({"note": "averages over the launches of each kernel template in one `bench.py --serial-streams` run under rocprofv3 --pmc; "
                   "see tools/summarize_pmc_util.py for the normalisation", "kernels": res},
          open(f"profiles/{tag}_pmc_mfma.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["cycles_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{k[:70]:70s} mfma_busy {v['mfma_busy']:.3f}  waitcnt {v['wave_cycles_waiting_on_waitcnt']:.2f}  cycles {v['cycles_per_launch']}")
