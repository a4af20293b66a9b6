import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(gemm_\w+_kernel<[^>]*>)", r["Kernel_Name"])
            if not m:
                continue
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
