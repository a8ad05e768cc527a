import csv, sys, glob, collections
for d in sys.argv[1:]:
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "")[:96]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[(k, row["Counter_Name"])] += 1
    for k, c in agg.items():
        if any(t in k for t in (sys.argv[0] and ["walk_joint", "walk_kernel", "walk_mixed", "walk_multi", "fftn_", "sep_conv", "direct_conv", "gmm_screen", "gmm_exact", "gmm_bucket", "gmm_bwd_max"])):
            print(d, k)
            for name, v in sorted(c.items()):
                print(f"   {name:28s} {v / n[(k, name)]:16.1f}  (x{n[(k, name)]})")
