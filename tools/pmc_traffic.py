"""Summarise two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, each collected alone with
--kernel-trace --output-format csv, as MI355X_MICROARCH.md §HBM prescribes) into per-kernel and
per-profiler-class HBM bytes per launch -> profiles/*_pmc_traffic.json (read by bench.py).

  python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<note>" [<mfma_counter_collection.csv> [<bench.json of a pass>]]

The optional third pass (--pmc MfmaUtil, the derived metric rocprofv3 -L documents:
100 * sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM)) adds the matrix-core busy
fraction per kernel (mean over its launches).

Units: the counters report KB.  gfx950 correction (guide): FETCH_SIZE counts 128-B requests as 64 B
for wide coalesced reads -> HBM read bytes ~= 2 * FETCH_SIZE * 1024; WRITE_SIZE is taken as is."""
import csv
import hashlib
import json
import os
import re
import sys
from collections import defaultdict


def csrc_sha16():
    """Hash of the kernel sources the profiled library was built from (bench.py compares it with the running build's)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "streamspeech_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]

def csrc_file_sha16():
    """Per-file hashes of the same sources: lets bench.py say WHICH files changed since the counter run."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "streamspeech_amd", "csrc")
    return {name: hashlib.sha256(open(os.path.join(d, name), "rb").read()).hexdigest()[:16]
            for name in sorted(os.listdir(d)) if name.endswith((".hip", ".hpp"))}


CLASSES = [("void ss::conv_sk2_kernel", "conv_sk2<256,128,32>"), ("void ss::conv_sk2_kernel", "conv_sk2_bf16x3<256,128,32>"), ("void ss::conv_sk_kernel", "conv_sk<128,BN,32>"), ("void ss::conv_slab_kernel<32", "conv_slab<32>"),
           ("void ss::conv_slab_kernel<16", "conv_slab<16>"),
           ("void ss::resblock_fused_kernel<32", "resblock_fused<32>"), ("void ss::resblock_fused_kernel<16", "resblock_fused<16>"),
           ("void ss::conv_gemm_kernel<32, 64, 32", "conv_gemm<32,64,32,2,2>"), ("void ss::conv_gemm_kernel<32, 32, 32", "conv_gemm<32,32,32,2,2>"),
           ("void ss::conv_gemm_kernel<128, 32, 32", "conv_gemm<128,32,32,4,1>"), ("void ss::conv_gemm_kernel<128, 16, 16", "conv_gemm<128,16,16,4,1>"),
           ("void ss::smallm_gemm_kernel<4, 1", "smallm_gemm<4,1>"), ("void ss::ffn_fused_kernel", "ffn_fused<256,2048>"),
           ("void ss::rt_linear_kernel", "rt_linear<48,256>"), ("void ss::rt_linear_kb_kernel", "rt_linear_kb<48,256>"), ("void ss::conv_c64_kernel", "conv_c64<256,64>"), ("void ss::conv_c64w_kernel", "conv_c64w<256,64>"), ("void ss::conv_c64w_kernel", "conv_c128w<256,128>"), ("void ss::conv_c64w_kernel", "conv_c32w<256,32>"), ("void ss::conv_c64w_kernel", "conv_c256w<256,128>"), ("void ss::conv_c32_kernel", "conv_c32<256,32>"), ("void ss::conv_c16_kernel", "conv_c16<256,16>")]


def read(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                tot[row["Kernel_Name"]] += float(row["Counter_Value"])
                n[row["Kernel_Name"]] += 1
    return tot, n


def main():
    fetch, nf = read(sys.argv[1], "FETCH_SIZE")
    write, nw = read(sys.argv[2], "WRITE_SIZE")
    kernels, classes = {}, {}
    for k in fetch:
        if not k.startswith(("void ss::", "ss::")) or nf[k] == 0:
            continue
        f_kb = fetch[k] / nf[k]
        w_kb = write.get(k, 0.0) / max(1, nw.get(k, 0))
        kernels[k] = {"launches": nf[k], "fetch_kb_per_launch": round(f_kb, 1), "write_kb_per_launch": round(w_kb, 1),
                      "hbm_mbytes_per_launch_corrected": round((2 * f_kb + w_kb) * 1024 / 1e6, 2)}
    def members(names, prefix, cls):      # the split-bf16 twins of conv_sk2 (third template argument true) are their own class
        ks = [k for k in names if k.startswith(prefix)]
        if prefix.endswith("conv_sk2_kernel"):
            x3 = cls.startswith("conv_sk2_bf16x3")
            ks = [k for k in ks if (re.search(r"conv_sk2_kernel<\d+, (true|false), true>", k) is not None) == x3]
        if prefix.endswith("conv_c64w_kernel"):   # one kernel template, two census classes: the channel count is its third template argument
            want = "256" if cls.startswith("conv_c256w") else "128" if cls.startswith("conv_c128w") else "32" if cls.startswith("conv_c32w") else "64"

            def channels(k):     # conv_c64w_kernel<DIL, CH, TAIL> (round 5) | <LRELU, DIL, CH> (round 4)
                m = re.search(r"conv_c64w_kernel<(\d+), (\d+), (\d+)>", k)
                if m:
                    return m.group(2)
                m = re.search(r"conv_c64w_kernel<(?:true|false), \d+, (\d+)>", k)
                return m.group(1) if m else None
            ks = [k for k in ks if channels(k) == want]
        return ks

    for prefix, cls in CLASSES:
        ks = members(kernels, prefix, cls)
        n = sum(kernels[k]["launches"] for k in ks)
        if n:
            classes[cls] = {"launches": n, "hbm_mbytes_per_launch_corrected": round(
                sum(kernels[k]["hbm_mbytes_per_launch_corrected"] * kernels[k]["launches"] for k in ks) / n, 2)}
    if len(sys.argv) > 5:      # third pass: rocprofv3 --pmc MfmaUtil (derived: MFMA busy cycles / (active cycles * SIMDs))
        util, nu = read(sys.argv[5], "MfmaUtil")
        for prefix, cls in CLASSES:
            ks = members(util, prefix, cls)
            n = sum(nu[k] for k in ks)
            if n and cls in classes:
                classes[cls]["mfma_util_pct"] = round(sum(util[k] for k in ks) / n, 1)
        for k in kernels:
            if nu.get(k, 0):
                kernels[k]["mfma_util_pct"] = round(util[k] / nu[k], 1)
    if len(sys.argv) > 6:      # bench.py's JSON line of one of the passes: its "process_census" counts the same launches
        census = json.load(open(sys.argv[6])).get("process_census", {})
        for cls, c in classes.items():
            z = census.get(cls)
            if z and z["launches"]:
                # the join is by KERNEL NAME (CLASSES maps a kernel-name prefix to the census class of that kernel's launcher), and it
                # only holds when the counter pass and the census saw the same launches of the same command
                c["census_launches"] = z["launches"]
                c["join_ok"] = z["launches"] == c["launches"]
                c["algo_mbytes_per_launch"] = round(z["algo_gbytes"] * 1e3 / z["launches"], 2)
                c["algo_gflop_per_launch"] = round(z["algo_tflop"] * 1e3 / z["launches"], 3)
                ratio = c["hbm_mbytes_per_launch_corrected"] / c["algo_mbytes_per_launch"]
                if c["join_ok"] and ratio >= 1.0:
                    c["traffic_over_algorithmic"] = round(ratio, 3)
                else:      # traffic below the algorithmic bytes is a join / census error (or Infinity-Cache-absorbed re-reads), never a measurement
                    c["traffic_over_algorithmic"] = None
                    c["traffic_over_algorithmic_refused"] = (f"raw ratio {ratio:.3f}; " + ("launch counts differ between the counter pass and the census"
                                                             if not c["join_ok"] else "below 1: the counters cannot see less than the compulsory bytes"))
    top = dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_mbytes_per_launch_corrected"] * kv[1]["launches"])[:64])
    json.dump({"note": sys.argv[4] if len(sys.argv) > 4 else "", "csrc_sha16": csrc_sha16(), "csrc_files_sha16": csrc_file_sha16(), "classes": classes, "kernels": top},
              open(sys.argv[3], "w"), indent=1)
    print(json.dumps(classes, indent=1))


if __name__ == "__main__":
    main()
