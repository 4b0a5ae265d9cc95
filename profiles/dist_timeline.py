#!/usr/bin/env python3
"""One step of a rocprofv3 --kernel-trace of `BLT_FORCE_DIST=1 python bench.py` (one rank through RCCL): where the weight-gradient flushes,
the gradient collectives (one-rank RCCL all-reduce = a device copy / RCCL kernel on the communication stream) and the optimiser fall
relative to the backward chain.  Usage: python profiles/dist_timeline.py <kernel_trace.csv> [step]"""
import csv, re, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I", n)
    return m.group(1) if m else n.split("(")[0][:60]


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if "prep_tokens" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 3
a, b = starts[k], starts[k + 1]
t0 = rows[a][0]
# queue of the collectives = the queue on which nccl kernels / the look-ahead conv stack run
print("step %d: %.3f ms, %d launches" % (k, (rows[b][0] - t0) / 1e6, b - a))
print("%10s %9s  %-5s %s" % ("start_us", "dur_us", "queue", "kernel"))
keep = ("wgrad_group", "nccl", "Nccl", "rccl", "adam_kernel", "sumsq", "ce_rows", "embed_scatter", "ln_param_reduce", "prep_tokens", "conv_stem_pool", "avgpool")
for s, e, n, q in rows[a:b]:
    if any(x in n for x in keep) or ("copyBuffer" in n and (e - s) > 20000):
        print("%10.1f %9.1f  q%-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
