import csv, sys, re
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n)
        m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I", n)
        n = m.group(1) if m else n.split("(")[0][:48]
        c, t = d.get(n, (0, 0.0)); d[n] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
steps = float(sys.argv[3])
rows = sorted(a, key=lambda n: -(b.get(n, (0, 0))[1] - a[n][1]))
print("%-50s %5s %9s %9s %9s" % ("kernel", "n/st", "A us/st", "B us/st", "delta"))
ta = tb = 0
for n in rows:
    ca, xa = a[n]; cb, xb = b.get(n, (0, 0.0))
    ta += xa; tb += xb
    if abs(xb - xa) / steps / 1e3 > 3: print("%-50s %5.1f %9.1f %9.1f %+9.1f" % (n, ca / steps, xa / steps / 1e3, xb / steps / 1e3, (xb - xa) / steps / 1e3))
print("total %.1f -> %.1f us/step" % (ta / steps / 1e3, tb / steps / 1e3))
