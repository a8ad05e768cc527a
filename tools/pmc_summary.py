import csv, sys, glob, collections
for d in sys.argv[1:]:
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:40]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[(k, row["Counter_Name"])] += 1
    for k, c in agg.items():
        if "sep_conv" in k or "direct_conv" in k:
            print(d, k)
            for name, v in sorted(c.items()):
                print(f"   {name:28s} {v / n[(k, name)]:16.1f}  (x{n[(k, name)]})")
