"""Per-kernel roofline table from a rocprofv3 kernel-stats CSV (rocprofv3 --stats --output-format csv, or tools/dbstats.py) and the PMC summary
(tools/pmc_traffic.py) of the SAME command: average duration, HBM-side GB/s (corrected FETCH+WRITE bytes /
duration) against 8 TB/s, MfmaUtil.   python tools/roofline_table.py <kernel_stats.csv> <pmc.json> > table.md"""
import csv
import json
import sys

stats = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
pmc = json.load(open(sys.argv[2]))["kernels"]
print("| kernel | launches | avg µs | % of kernel time | HBM-side MB/launch | GB/s | frac of 8 TB/s | MfmaUtil % |")
print("|---|---|---|---|---|---|---|---|")
for name, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["Percentage"]))[:14]:
    p = pmc.get(name)
    us = float(r["AverageUs"]) if "AverageUs" in r else float(r["AverageNs"]) / 1e3    # rocprofv3 kernel_stats.csv reports ns, tools/dbstats.py us
    if p:
        mb = p["hbm_mbytes_per_launch_corrected"]
        gbs = mb / us * 1e3
        print(f"| `{name.split('(')[0].replace('void ss::', '')}` | {r['Calls']} | {us:.1f} | {float(r['Percentage']):.1f} | {mb:.1f} | {gbs:.0f} | "
              f"{gbs / 8000:.2f} | {p.get('mfma_util_pct', '—')} |")
    else:
        print(f"| `{name.split('(')[0].replace('void ss::', '')}` | {r['Calls']} | {us:.1f} | {float(r['Percentage']):.1f} | — | — | — | — |")
