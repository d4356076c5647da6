import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:100]
    e = agg[k][r["Counter_Name"]]
    e[0] += 1; e[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print("   %-28s %14.0f per launch (%d launches)" % (c, v / n, n))
