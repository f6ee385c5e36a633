"""Per-kernel mean of each PMC counter from a rocprofv3 --pmc CSV directory."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:70]
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc, key=lambda k: -sum(v[0] for v in acc[k].values())):
    print(k)
    for c, (s, n) in sorted(acc[k].items()):
        print("    %-32s mean %.4g  (n=%d)" % (c, s / n, n))
