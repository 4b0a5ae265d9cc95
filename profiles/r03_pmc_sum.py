import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0]
            if "gemm_nt2" not in n: continue
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in agg.items():
    print(n)
    wc = sum(c.get("SQ_WAVE_CYCLES", [0])) / max(len(c.get("SQ_WAVE_CYCLES", [1])), 1)
    for k, v in sorted(c.items()):
        m = sum(v) / len(v)
        print("   %-30s %14.0f   %s" % (k, m, ("%.3f of wave-cycles" % (m / wc)) if wc and k.startswith("SQ_") else ""))
