"""Summarise rocprofv3 --pmc CSVs: mean counter value per launch of kernels whose name contains argv[2]."""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(d + "/p*/p_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-36s n=%3d mean=%.4g" % (k, len(v), sum(v) / len(v)))
