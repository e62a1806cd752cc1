"""Per-kernel averages of rocprofv3 --pmc CSV output.  usage: pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "gemm_kernel"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + "/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k:
            continue
        acc[k.split("(")[0][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s avg %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
