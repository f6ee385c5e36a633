"""Per-kernel mean / max of each PMC counter from a rocprofv3 --pmc CSV directory."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:78]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc, key=lambda k: -sum(sum(v) for v in acc[k].values())):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("    %-32s mean %.5g  max %.5g  (n=%d)" % (c, sum(v) / len(v), max(v), len(v)))
