"""Summarise rocprofv3 --pmc CSV output: per-kernel mean of each counter."""
import csv, sys, glob, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void hank::", "").replace("hank::", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    n = max(len(v) for v in acc[k].values())
    print(f"{k}  (dispatches {n})")
    for c, v in sorted(acc[k].items()):
        print(f"    {c:28s} mean {sum(v)/len(v):16.1f}")
