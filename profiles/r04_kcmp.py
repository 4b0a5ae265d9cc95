"""compare two kernel traces (per step, by kernel family): python scratch/kcmp4.py a.csv b.csv"""
import csv, re, sys
from collections import defaultdict
def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I", name)
    if m: return m.group(1)
    return re.sub(r", 163840>", ">", name.split("(")[0])[:64]
def load(p):
    rows = []
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    st = [i for i, r in enumerate(rows) if "prep_tokens" in r[2]]
    a, b = st[2], st[-1]
    n = len(st) - 3
    fam = defaultdict(lambda: [0, 0.0])
    for s, e, k in rows[a:b]:
        f = fam[short(k)]; f[0] += 1; f[1] += (e - s) / 1e3
    return {k: (v[0] / n, v[1] / n) for k, v in fam.items()}, (rows[b][0] - rows[a][0]) / 1e6 / n
A, wa = load(sys.argv[1]); B, wb = load(sys.argv[2])
print("wall per step: %.3f vs %.3f ms" % (wa, wb))
keys = sorted(set(A) | set(B), key=lambda k: -max(A.get(k, (0, 0))[1], B.get(k, (0, 0))[1]))
ta = tb = 0
for k in keys[:45]:
    a, b = A.get(k, (0, 0)), B.get(k, (0, 0)); ta += a[1]; tb += b[1]
    print("%-60s %5.1f x %6.1f = %7.1f | %5.1f x %6.1f = %7.1f  (%+.1f)" % (k, a[0], a[1] / max(a[0], 1e-9), a[1], b[0], b[1] / max(b[0], 1e-9), b[1], b[1] - a[1]))
print("sum of listed: %.1f vs %.1f us" % (ta, tb))
