"""Print per-kernel averages of every counter found under the given rocprofv3 --pmc output dirs."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "svae::" in k and "split_weights" not in k:
                acc[k.split("(")[0].replace("void svae::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    print("   " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
