"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel-name average of each counter."""
import csv, glob, sys, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in acc.items():
    out[k] = {c: {"launches": len(x), "avg": sum(x) / len(x)} for c, x in v.items()}
print(json.dumps(out, indent=1))
