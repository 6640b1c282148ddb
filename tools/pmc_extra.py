"""Per-kernel means of further rocprofv3 counters / derived metrics (one --pmc pass each, --kernel-trace only) as a markdown table:
  python tools/pmc_extra.py <out.md> <name>=<counter_collection.csv> [...]
Rows = the kernels that take >= 1 % of the launches' summed value of the first metric's pass duration proxy (launch count x mean), i.e.
every kernel class the share table names; values are means over a kernel's launches (a derived metric such as LdsBankConflict or
VALUBusy is already a percentage per launch)."""
import csv
import re
import sys
from collections import defaultdict

KEEP = ("conv_sk2_kernel", "conv_c64_kernel", "conv_c64w_kernel", "conv_c32_kernel", "conv_c16_kernel", "resblock_fused_kernel", "ffn_fused_kernel",
        "rt_linear_kernel", "conv_gemm_kernel", "smallm_gemm_kernel", "attention_relpos_mfma_kernel", "attention_decode_kernel",
        "conv_slab_kernel", "fbank", "dwconv")


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("ss::", "")


def main():
    out = sys.argv[1]
    cols, table, launches = [], defaultdict(dict), defaultdict(int)
    for spec in sys.argv[2:]:
        metric, path = spec.split("=", 1)
        cols.append(metric)
        tot, n = defaultdict(float), defaultdict(int)
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == metric:
                    tot[row["Kernel_Name"]] += float(row["Counter_Value"])
                    n[row["Kernel_Name"]] += 1
        for k in tot:
            table[short(k)][metric] = tot[k] / n[k]
            launches[short(k)] = max(launches[short(k)], n[k])
    rows = [k for k in table if any(s in k for s in KEEP) and launches[k] >= 8]
    rows.sort(key=lambda k: -launches[k])
    with open(out, "w") as f:
        f.write("| kernel | launches | " + " | ".join(cols) + " |\n|---|---|" + "---|" * len(cols) + "\n")
        for k in rows:
            f.write(f"| `{k}` | {launches[k]} | " + " | ".join(f"{table[k][c]:.2f}" if c in table[k] else "—" for c in cols) + " |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
