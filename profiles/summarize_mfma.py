#!/usr/bin/env python3
"""Summarises a rocprofv3 PMC pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES and SQ_INSTS_VALU_MFMA_MOPS_BF16:
   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv \\
             -d gpurun_out/fin_mfma -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
per GEMM-shaped kernel.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 MFMA pipes (256 CUs x 4 SIMDs): a
v_mfma_f32_16x16x32_bf16 holds its pipe for 16 cycles, and MOPS_BF16 counts 512 flops per unit (checked: MOPS / 32 x 16 = BUSY for
the conv kernels, MOPS x 512 = the padded-pitch flop count).  mfma_util_wall = BUSY / (1024 x duration x 2.4 GHz): against the same
2.4 GHz peak the roofline in bench.py uses.  Usage: summarize_mfma.py <pmc_dir> <out.csv>"""
import collections, csv, glob, sys


def family(name):
    n = name.replace("(anonymous namespace)::", "")
    if "conv3x3_pp_kernel" in n or "conv_stem_direct" in n or "gemm_dma_kernel" in n:
        return n.split("(")[0].replace("void ", "")
    if "gemm_kernel" in n:
        return "gemm_kernel (register-staged: weight gradients, fp32 head)"
    return None


def main():
    src, out = sys.argv[1:3]
    f = glob.glob(src + "/*/*_counter_collection.csv")[0]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = family(r["Kernel_Name"])
        if k is None:
            continue
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(out, "w") as fh:
        fh.write("kernel,launches,avg_duration_us,mfma_busy_cycles,busy_cu_cycles,mfma_flops_G,mfma_util_wall,mfma_util_of_cu_busy\n")
        for k in sorted(vals):
            v = vals[k]
            n = len(v["SQ_BUSY_CU_CYCLES"])
            busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / n
            cu = sum(v["SQ_BUSY_CU_CYCLES"]) / n
            mops = sum(v["SQ_INSTS_VALU_MFMA_MOPS_BF16"]) / n
            d = sum(dur[k]) / n
            fh.write("%s,%d,%.2f,%.0f,%.0f,%.2f,%.4f,%.4f\n" % (k.replace(",", ";"), n, d / 1e3, busy, cu, mops * 512 / 1e9,
                                                               busy / (1024 * d * 2.4), busy / (4 * cu) if cu else 0.0))
    print(open(out).read())


if __name__ == "__main__":
    main()
