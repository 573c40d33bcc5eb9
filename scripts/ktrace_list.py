"""Per-launch durations of kernels matching argv[2] from a rocprofv3 kernel-trace CSV, in launch order."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
out = {}
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if sys.argv[2] in n:
        out.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in out.items():
    print(n, " ".join("%.1f" % x for x in v))
