"""Sum rocprofv3 --pmc counter_collection.csv per kernel name and counter."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            import re
            m = re.search(r"(\w+_kernel(?:<\w+>)?)", k)
            k = m.group(1) if m else k[:28]
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for (k, c), v in sorted(acc.items()):
            if True:
                print(f"{k:16s} {c:28s} {v:18.0f}  dispatches {n[(k,c)]}")
