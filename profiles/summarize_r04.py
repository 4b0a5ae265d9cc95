#!/usr/bin/env python3
"""Round-4 profile summaries from profiles/collect_r04.sh's outputs (gpurun_out/r04_{stats,fetch,write,mfma}):
  r04_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench (copied)
  r04_pmc_kernels.csv          per kernel: launches, HBM read / write MB per launch (reads = 2 x FETCH_SIZE: gfx950 wide-load correction)
  r04_pmc_traffic.json         per-launch HBM bytes of the two kernel families bench.py reports a roofline for (it reads this file)
  r04_pmc_mfma_busy.csv        per kernel family: MFMA-pipe busy share (SQ_VALU_MFMA_BUSY_CYCLES / (1024 pipes x duration x 2.4 GHz))
Usage: python profiles/summarize_r04.py [gpurun_out]"""
import collections, csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.abspath(__file__))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(ROOT), "gpurun_out")


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I", n)
    return m.group(1) if m else n.split("(")[0]


def is_gemm(n):       # the Linear-layer GEMM family (bench.py class 1)
    s = short(n)
    return s.startswith("gemm_nt2_kernel") or s.startswith("wgrad_group_kernel") or re.match(r"gemm_dma_kernel<\d+, \d+, 0, \d+>", s) is not None or \
        (s == "gemm_kernel")


def is_conv(n):
    s = short(n)
    return s.startswith("conv3x3_pp_kernel") or s.startswith("conv_stem_direct_kernel") or s.startswith("conv_stem_pool_kernel") or re.match(r"gemm_dma_kernel<\d+, \d+, [12], \d+>", s) is not None


def counters(d):
    fs = glob.glob(os.path.join(SRC, d, "*", "*_counter_collection.csv"))
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    if not fs:
        return out, dur
    seen = set()
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (r.get("Dispatch_Id"), r["Kernel_Name"])
        if key not in seen and "Start_Timestamp" in r:
            seen.add(key)
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return out, dur


def main():
    st = glob.glob(os.path.join(SRC, "r04_stats", "*", "*_kernel_stats.csv"))
    if st:
        shutil.copy(max(st, key=os.path.getmtime), os.path.join(ROOT, "r04_bench_kernel_stats.csv"))
    F, _ = counters("r04_fetch")
    W, _ = counters("r04_write")
    rows = []
    for k in sorted(F, key=lambda k: -sum(F[k]["FETCH_SIZE"])):
        n = len(F[k]["FETCH_SIZE"])
        rd = 2.0 * sum(F[k]["FETCH_SIZE"]) / n / 1024.0
        w = W.get(k, {}).get("WRITE_SIZE", [0.0])
        rows.append((short(k), n, rd, sum(w) / max(1, len(w)) / 1024.0, k))
    with open(os.path.join(ROOT, "r04_pmc_kernels.csv"), "w") as fh:
        fh.write("kernel,launches,read_MB_per_launch(2xFETCH_SIZE),write_MB_per_launch\n")
        for r in rows:
            fh.write('"%s",%d,%.3f,%.3f\n' % r[:4])

    def fam(pred):
        sel = [r for r in rows if pred(r[4])]
        n = sum(r[1] for r in sel)
        if not n:
            return None
        rd = sum(r[2] * r[1] for r in sel) / n
        wr = sum(r[3] * r[1] for r in sel) / n
        return {"launches_sampled": n, "read_MB_per_launch": round(rd, 2), "write_MB_per_launch": round(wr, 2), "bytes_per_launch": int((rd + wr) * 1e6)}
    g, c = fam(is_gemm), fam(is_conv)
    # average launch duration of each family in the rocprofv3 --kernel-trace --stats run of the same bench command
    ks = os.path.join(ROOT, "r04_bench_kernel_stats.csv")
    if os.path.exists(ks):
        kr = list(csv.DictReader(open(ks)))
        for d, pred in ((g, is_gemm), (c, is_conv)):
            sel = [r for r in kr if pred(r["Name"])]
            calls = sum(int(r["Calls"]) for r in sel)
            if d is not None and calls:
                d["rocprof_avg_launch_us"] = round(sum(int(r["TotalDurationNs"]) for r in sel) / calls / 1e3, 2)
                d["rocprof_launches"] = calls
    import subprocess
    try:
        commit = subprocess.run(["git", "-C", os.path.dirname(ROOT), "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        commit = ""
    tr = {"collected_by": "collect_r04.sh", "commit": commit, "note": "HBM bytes per launch, big cfg B=256 bf16: reads = 2 x FETCH_SIZE (gfx950 wide-load correction), writes = WRITE_SIZE; separate --pmc passes "
                  "(profiles/collect_r04.sh)", "gemm": g, "conv": c}
    if g:
        tr["gemm_bytes_per_launch"] = g["bytes_per_launch"]
    if c:
        tr["conv_bytes_per_launch"] = c["bytes_per_launch"]
    json.dump(tr, open(os.path.join(ROOT, "r04_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(tr, indent=1))
    M, D = counters("r04_mfma")
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
    for k, v in M.items():
        if not (is_gemm(k) or is_conv(k)):
            continue
        a = agg[short(k)]
        n = len(v["SQ_VALU_MFMA_BUSY_CYCLES"])
        a[0] += n
        a[1] += sum(D[k])
        a[2] += sum(v["SQ_VALU_MFMA_BUSY_CYCLES"])
        a[3] += sum(v["SQ_BUSY_CU_CYCLES"])
        a[4] += sum(v["SQ_INSTS_VALU_MFMA_MOPS_BF16"])
    with open(os.path.join(ROOT, "r04_pmc_mfma_busy.csv"), "w") as fh:
        fh.write("kernel,launches,avg_duration_us,mfma_flops_G_per_launch,mfma_util_wall(2.4GHz),mfma_util_of_cu_busy\n")
        tot = [0.0, 0.0]
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            fh.write('"%s",%d,%.2f,%.2f,%.4f,%.4f\n' % (k, a[0], a[1] / a[0] / 1e3, a[4] * 512 / a[0] / 1e9, a[2] / (1024 * a[1] * 2.4), a[2] / (4 * a[3]) if a[3] else 0))
            if k.startswith("gemm_nt2") or k.startswith("wgrad_group") or k.startswith("gemm_kernel") or re.match(r"gemm_dma_kernel<\d+, \d+, 0", k):
                tot[0] += a[2]; tot[1] += a[1]
        if tot[1]:
            fh.write('"Linear-GEMM family (time-weighted)",,,,%.4f,\n' % (tot[0] / (1024 * tot[1] * 2.4)))
    print(open(os.path.join(ROOT, "r04_pmc_mfma_busy.csv")).read())
    # vector-memory path (texture addresser = the L1 / L2 -> LDS intake of a CU): busy share per kernel next to the MFMA share
    if not glob.glob(os.path.join(SRC, "r04_ta", "*", "*_counter_collection.csv")):
        return
    T, DT = counters("r04_ta")
    with open(os.path.join(ROOT, "r04_pmc_ta_busy.csv"), "w") as fh:
        fh.write("kernel,launches,avg_duration_us,ta_busy_avr_share_of_wall(2.4GHz),ta_busy_max_share,tcp_pending_stall_share_per_tcp(256 TCPs)\n")
        agg2 = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
        for k, v in T.items():
            if not (is_gemm(k) or is_conv(k) or "layernorm" in k or "attn_" in k or "adam" in k):
                continue
            a = agg2[short(k)]
            n = len(v.get("TA_BUSY_avr", []))
            a[0] += n; a[1] += sum(DT[k]); a[2] += sum(v.get("TA_BUSY_avr", [])); a[3] += sum(v.get("TA_BUSY_max", [])); a[4] += sum(v.get("TCP_PENDING_STALL_CYCLES_sum", []))
        for k, a in sorted(agg2.items(), key=lambda kv: -kv[1][1]):
            if a[0]:
                cyc = a[1] * 2.4      # kernel cycles at 2.4 GHz (durations are ns), summed over the launches
                fh.write('"%s",%d,%.2f,%.3f,%.3f,%.3f\n' % (k, a[0], a[1] / a[0] / 1e3, a[2] / cyc, a[3] / cyc, a[4] / (256.0 * cyc)))
    print(open(os.path.join(ROOT, "r04_pmc_ta_busy.csv")).read())


if __name__ == "__main__":
    main()
