"""A/B of two rocprofv3 --kernel-trace --stats runs: per-kernel total time difference (ms), sorted by |delta|.
usage: ab_kernels.py <dirA> <dirB> [n_steps_counted]"""
import csv, glob, sys, collections
def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    t = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        t[r["Name"]] = (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]))
    return t
a, b = load(sys.argv[1]), load(sys.argv[2])
n = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = []
for k in set(a) | set(b):
    ta, ca = a.get(k, (0.0, 0)); tb, cb = b.get(k, (0.0, 0))
    rows.append((tb - ta, k, ta, ca, tb, cb))
rows.sort(key=lambda r: -abs(r[0]))
print(f"total A {sum(v[0] for v in a.values())/n:.2f} ms/step, B {sum(v[0] for v in b.values())/n:.2f} ms/step")
for d, k, ta, ca, tb, cb in rows[:30]:
    print(f"{d/n:+8.3f} ms/step  A {ta/n:8.3f} ({ca:5d})  B {tb/n:8.3f} ({cb:5d})  {k[:110]}")
