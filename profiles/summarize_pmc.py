#!/usr/bin/env python3
"""Summarises two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately as the MI355X guide prescribes:
   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline)
into per-kernel HBM traffic.  gfx950 correction: FETCH_SIZE under-reports wide (16 B/lane) coalesced reads by exactly 2x, so
reads = 2 * FETCH_SIZE KB; WRITE_SIZE is exact.  Usage: summarize_pmc.py <fetch_dir> <write_dir> <out_prefix>"""
import collections, csv, glob, json, sys


def load(path, cname):
    f = glob.glob(path + "/*/*_counter_collection.csv")[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cname:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


def main():
    fetch, write, out = sys.argv[1:4]
    F, W = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    rows = []
    for k in sorted(F, key=lambda k: -sum(F[k])):
        n = len(F[k])
        rd = 2.0 * sum(F[k]) / n / 1024.0
        wr = sum(W.get(k, [0.0])) / max(1, len(W.get(k, [0.0]))) / 1024.0
        rows.append((k.replace("(anonymous namespace)::", ""), n, rd, wr))
    with open(out + "_kernels.csv", "w") as fh:
        fh.write("kernel,launches,read_MB_per_launch(2xFETCH_SIZE),write_MB_per_launch\n")
        for r in rows:
            fh.write('"%s",%d,%.3f,%.3f\n' % r)
    # the convolution launches of the ResNet-18 stack: LDS-patch 3x3 kernel, direct stem kernel, implicit-GEMM kernel with the conv (1) or
    # stem (2) loader (template arguments <BM, BN, LOADER, NST>)
    import re
    conv = [k for k in F if "conv3x3_pp_kernel" in k or "conv_stem_direct_kernel" in k or re.search(r"gemm_dma_kernel<\d+, \d+, [12], \d+>", k)]
    n = sum(len(F[k]) for k in conv)
    rd = 2.0 * sum(sum(F[k]) for k in conv) / n / 1024.0
    wr = sum(sum(W[k]) for k in conv if k in W) / max(1, sum(len(W[k]) for k in conv if k in W)) / 1024.0
    json.dump({"kernel": "conv3x3_pp_kernel + conv_stem_direct_kernel + gemm_dma_kernel<*,*,conv,*>", "launches_sampled": n, "read_MB_per_launch": round(rd, 2),
               "write_MB_per_launch": round(wr, 2), "traffic_MB_per_launch": round(rd + wr, 2),
               "note": "reads = 2 x FETCH_SIZE (gfx950 wide-load correction), writes = WRITE_SIZE; separate --pmc passes"},
              open(out + "_conv_traffic.json", "w"), indent=1)
    print(open(out + "_conv_traffic.json").read())


if __name__ == "__main__":
    main()
