"""Timeline statistics from a rocprofv3 --kernel-trace CSV: busy union, idle gaps, concurrency, per-kernel sums
over the last `nsteps` occurrences of the step marker kernel (sgd_kernel)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
sgd = [e for e in ev if "sgd_kernel" in e[2]]
t0, t1 = sgd[-4][1], sgd[-1][1]          # three full steps
sel = [e for e in ev if e[0] >= t0 and e[1] <= t1]
tot = sum(e[1] - e[0] for e in sel)
pts = []
for s, e, _ in sel:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = 0; two = 0; depth = 0; last = None
for t, d in pts:
    if depth >= 1: busy += t - last
    if depth >= 2: two += t - last
    depth += d; last = t
wall = t1 - t0
print("steps 3: wall %.2f ms/step, busy(union) %.2f, idle %.2f, >=2 kernels %.2f, sum of kernel time %.2f" %
      (wall / 3e6, busy / 3e6, (wall - busy) / 3e6, two / 3e6, tot / 3e6))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in sel:
    k = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    agg[k][0] += e - s; agg[k][1] += 1
for k, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:22]:
    print("%-62s %8.3f ms/step %6d calls/step avg %.1f us" % (k, d / 3e6, c // 3, d / c / 1e3))

# classes: contraction kernels (MFMA-bound) vs everything else (HBM- / latency-bound)
def is_mfma(n):
    return any(t in n for t in ("conv_igemm", "conv_wgrad", "head_sweep"))
pts = []
for s, e, n in sel:
    c = 0 if is_mfma(n) else 1
    pts.append((s, c, 1)); pts.append((e, c, -1))
pts.sort()
d = [0, 0]; last = None
t_m = t_o = t_both = t_m2 = 0
for t, c, dd in pts:
    if last is not None:
        dt = t - last
        if d[0] and d[1]: t_both += dt
        elif d[0]: t_m += dt
        elif d[1]: t_o += dt
        if d[0] >= 2: t_m2 += dt
    d[c] += dd; last = t
print("per step: contraction kernels only %.2f ms, contraction + other %.2f, other only %.2f, idle %.2f; >= 2 contraction kernels %.2f" %
      (t_m / 3e6, t_both / 3e6, t_o / 3e6, (wall - t_m - t_both - t_o) / 3e6, t_m2 / 3e6))
