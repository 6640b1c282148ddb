"""Kernel-time share against algorithmic-FLOP share per kernel class of ONE process: a rocprofv3 kernel-stats CSV and the
`process_census` of the bench.py line that process printed (the census counts the algorithmic FLOPs of exactly the launches
the profile saw).  A class whose time share is far above its FLOP share is the one running below the others' efficiency.
    python tools/share_table.py <kernel_stats.csv> <bench.json> > table.md"""
import csv
import json
import re
import sys

# profiler class (bench.py process_census key) -> kernel-name prefixes of the profile
CLASSES = [("conv_sk2<256,128,32>", ["void ss::conv_sk2_kernel"]),
           ("conv_sk<128,BN,32>", ["void ss::conv_sk_kernel"]),
           ("conv_c64<256,64>", ["void ss::conv_c64_kernel"]),
           # one kernel template (csrc/conv_c64w.hip), four census classes: the channel count is its SECOND template argument
           # (conv_c64w_kernel<DIL, CH, TAIL> since round 5; <LRELU, DIL, CH> in round 4: "re:" entries are regular expressions)
           ("conv_c64w<256,64>", [r"re:void ss::conv_c64w_kernel<(?:\d+, 64, \d+|(?:true|false), \d+, 64)>"]),
           ("conv_c32w<256,32>", [r"re:void ss::conv_c64w_kernel<(?:\d+, 32, \d+|(?:true|false), \d+, 32)>"]),
           ("conv_c128w<256,128>", [r"re:void ss::conv_c64w_kernel<(?:\d+, 128, \d+|(?:true|false), \d+, 128)>"]),
           ("conv_c256w<256,128>", [r"re:void ss::conv_c64w_kernel<\d+, 256, \d+>"]),
           ("conv_c32<256,32>", ["void ss::conv_c32_kernel"]),
           ("conv_c16<256,16>", ["void ss::conv_c16_kernel"]),
           ("resblock_fused<32>", ["void ss::resblock_fused_kernel<32"]),
           ("resblock_fused<16>", ["void ss::resblock_fused_kernel<16"]),
           ("conv_slab<32>", ["void ss::conv_slab_kernel<32", "void ss::conv_pair_kernel<32"]),
           ("conv_slab<16>", ["void ss::conv_slab_kernel<16", "void ss::conv_pair_kernel<16"]),
           ("ffn_fused<256,2048>", ["void ss::ffn_fused_kernel"]),
           ("rt_linear<48,256>", ["void ss::rt_linear_kernel"]),
           ("rt_linear_kb<48,256>", ["void ss::rt_linear_kb_kernel"]),
           ("conv_gemm<32,64,32,2,2>", ["void ss::conv_gemm_kernel<32, 64, 32"]),
           ("conv_gemm<32,32,32,2,2>", ["void ss::conv_gemm_kernel<32, 32, 32"]),
           ("conv_gemm<32,64,16,2,2>", ["void ss::conv_gemm_kernel<32, 64, 16"]),
           ("conv_gemm<128,32,32,4,1>", ["void ss::conv_gemm_kernel<128, 32, 32"]),
           ("conv_gemm<128,16,16,4,1>", ["void ss::conv_gemm_kernel<128, 16, 16"]),
           ("smallm_gemm<4,1>", ["void ss::smallm_gemm_kernel<4, 1"]),
           ("smallm_gemm<2,2>", ["void ss::smallm_gemm_kernel<2, 2"]),
           ("smallm_gemm<1,4>", ["void ss::smallm_gemm_kernel<1, 4"])]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    census = json.load(open(sys.argv[2]))["process_census"]
    total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    total_tf = sum(c["algo_tflop"] for c in census.values())
    print("| kernel class | launches | kernel time ms | share of kernel time | algorithmic TFLOP | share of FLOPs | TFLOP/s | of 157.3 |")
    print("|---|---|---|---|---|---|---|---|")
    seen = 0.0
    for cls, prefixes in CLASSES:
        rs = [r for r in rows if any(re.match(p[3:], r["Name"]) if p.startswith("re:") else r["Name"].startswith(p) for p in prefixes)]
        if not rs or cls not in census:
            continue
        ns = sum(float(r["TotalDurationNs"]) for r in rs)
        seen += ns
        tf = census[cls]["algo_tflop"]
        print(f"| `{cls}` | {sum(int(r['Calls']) for r in rs)} | {ns / 1e6:.1f} | {100 * ns / total_ns:.1f} % | {tf:.2f} | {100 * tf / total_tf:.1f} % | "
              f"{tf / (ns * 1e-9):.1f} | {tf / (ns * 1e-9) / 157.3:.2f} |")
    print(f"| everything else (attention, LayerNorm, depthwise conv, argmax, fbank, GEMVs, copies) | | {(total_ns - seen) / 1e6:.1f} | "
          f"{100 * (total_ns - seen) / total_ns:.1f} % | | | | |")
    print(f"| **all kernels** | | {total_ns / 1e6:.1f} | 100 % | {total_tf:.2f} | 100 % | {total_tf / (total_ns * 1e-9):.1f} | {total_tf / (total_ns * 1e-9) / 157.3:.2f} |")


if __name__ == "__main__":
    main()
