"""Which kernels precede / follow launches of a given kernel in a rocprofv3 kernel trace (who issues those small copies?)."""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:50] for r in rows]
prev, nxt = collections.Counter(), collections.Counter()
for i, n in enumerate(names):
    if sys.argv[2] in n:
        prev[names[i - 1] if i else "-"] += 1
        nxt[names[i + 1] if i + 1 < len(names) else "-"] += 1
print("before:", prev.most_common(8))
print("after: ", nxt.most_common(8))
