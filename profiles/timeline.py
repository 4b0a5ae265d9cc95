#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV: per step (delimited by the prep_tokens launch) the busy time per queue, the
critical-path gaps and a per-kernel-family summary.  Usage: python profiles/timeline.py <kernel_trace.csv> [step_index]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I", name)
    if m:
        return m.group(1)
    return name.split("(")[0][:60]


def main():
    rows = []
    with open(sys.argv[1]) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0"),
                         int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "prep_tokens" in r[2]]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 2
    a, b = starts[which], starts[which + 1]
    step = rows[a:b]
    t0 = step[0][0]
    print("step %d: %d launches, wall %.3f ms" % (which, len(step), (rows[b][0] - t0) / 1e6))
    fam = defaultdict(lambda: [0, 0.0])
    for s, e, n, q, st, g, w in step:
        f = fam[short(n)]
        f[0] += 1
        f[1] += (e - s) / 1e3
    tot = sum(v[1] for v in fam.values())
    print("kernel time sum %.3f ms" % (tot / 1e3))
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:28]:
        print("  %-46s %4d x %8.1f us = %8.1f us (%4.1f%%)" % (k, v[0], v[1] / v[0], v[1], 100 * v[1] / tot))
    # union busy time and idle gaps
    ev = sorted((s, e) for s, e, *_ in step)
    busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
    for s, e in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e - t0))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("GPU busy (any kernel) %.3f ms, idle %.3f ms in %d gaps (mean %.2f us)" % (busy / 1e6, sum(g for g, _ in gaps) / 1e6, len(gaps),
                                                                                  sum(g for g, _ in gaps) / max(len(gaps), 1) / 1e3))
    if len(sys.argv) > 3:
        for s, e, n, q, st, g, w in step:
            print("%9.1f %8.1f q%-3s s%-3s %-40s grid %d" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, short(n), g // max(w, 1)))


if __name__ == "__main__":
    main()
