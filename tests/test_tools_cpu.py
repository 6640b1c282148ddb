"""The profiling helpers under tools/ that produce the numbers DESIGN.md and bench.py cite (profiles/r02_pmc_traffic.json,
profiles/r02_trace_gaps_*.txt, profiles/r02_v27_roofline_table.md), run on tiny synthetic rocprofv3 outputs."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SK2 = "void ss::conv_sk2_kernel<128, false, false>(ss::GemmArgs, ss::Sk2Args)"
SK2_X3 = "void ss::conv_sk2_kernel<128, false, true>(ss::GemmArgs, ss::Sk2Args)"
SLAB = "void ss::conv_slab_kernel<32, 32, true>(ss::GemmArgs, int)"


def _run(*argv):
    return subprocess.run([sys.executable, *argv], cwd=ROOT, capture_output=True, text=True, check=True).stdout


def _counter_csv(path, counter, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
        for name, value in rows:
            w.writerow([name, counter, value])


def test_pmc_traffic_classes_corrections_and_census(tmp_path):
    fetch, write, util, out, bench = (str(tmp_path / n) for n in ("f.csv", "w.csv", "u.csv", "o.json", "b.json"))
    _counter_csv(fetch, "FETCH_SIZE", [(SK2, 100000.0), (SK2, 120000.0), (SK2_X3, 50000.0), (SLAB, 40000.0)])      # KB
    _counter_csv(write, "WRITE_SIZE", [(SK2, 90000.0), (SK2, 110000.0), (SK2_X3, 60000.0), (SLAB, 30000.0)])
    _counter_csv(util, "MfmaUtil", [(SK2, 60.0), (SK2, 64.0), (SK2_X3, 30.0), (SLAB, 44.0)])
    json.dump({"process_census": {"conv_sk2<256,128,32>": {"launches": 2, "algo_tflop": 0.07, "algo_gbytes": 0.5},
                                  "conv_slab<32>": {"launches": 1, "algo_tflop": 0.02, "algo_gbytes": 0.1}}}, open(bench, "w"))
    _run("tools/pmc_traffic.py", fetch, write, out, "note", util, bench)
    d = json.load(open(out))
    sk2 = d["classes"]["conv_sk2<256,128,32>"]
    # two f32 launches: HBM bytes = 2 * FETCH (gfx950 correction) + WRITE, in MB with KB = 1024 B
    want = ((2 * 110000.0 + 100000.0) * 1024) / 1e6
    assert sk2["launches"] == 2 and abs(sk2["hbm_mbytes_per_launch_corrected"] - want) < 0.01
    assert abs(sk2["mfma_util_pct"] - 62.0) < 1e-6
    assert abs(sk2["algo_mbytes_per_launch"] - 250.0) < 1e-6 and abs(sk2["traffic_over_algorithmic"] - want / 250.0) < 1e-3
    x3 = d["classes"]["conv_sk2_bf16x3<256,128,32>"]          # the split-bf16 twin is its own class
    assert x3["launches"] == 1 and abs(x3["mfma_util_pct"] - 30.0) < 1e-6
    assert d["classes"]["conv_slab<32>"]["launches"] == 1


def test_trace_gaps_busy_fraction_and_window(tmp_path):
    trace, bench = str(tmp_path / "t.csv"), str(tmp_path / "b.json")
    with open(trace, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id"])
        w.writerow(["warm(x)", 0, 1000000, 1])                        # outside the timed region
        w.writerow(["big(x)", 2000000, 2600000, 1])                   # 0.6 ms
        w.writerow(["small(x)", 2500000, 2510000, 2])                 # 10 us inside `big`
        w.writerow(["small(x)", 2800000, 2810000, 2])                 # 10 us alone
    json.dump({"timed_region_monotonic_ns": [2000000, 3000000]}, open(bench, "w"))
    out = _run("tools/trace_gaps.py", trace, bench)
    first = out.splitlines()[0]
    assert "window ms 1.0" in first and "busy 0.610" in first and "idle 0.390" in first and "only-short-kernels 0.010" in first
    assert "hw queues 2" in first


def test_roofline_table_reads_rocprofv3_kernel_stats(tmp_path):
    stats, pmc = str(tmp_path / "s.csv"), str(tmp_path / "p.json")
    with open(stats, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        w.writerow([SK2, 10, 3000000, 300000.0, 75.0, 1, 2, 0.0])
        w.writerow([SLAB, 5, 1000000, 200000.0, 25.0, 1, 2, 0.0])
    json.dump({"kernels": {SK2: {"hbm_mbytes_per_launch_corrected": 300.0, "mfma_util_pct": 62.0}}}, open(pmc, "w"))
    out = _run("tools/roofline_table.py", stats, pmc)
    rows = [r for r in out.splitlines() if r.startswith("| `")]
    assert len(rows) == 2
    assert "conv_sk2_kernel<128, false, false>" in rows[0] and "| 10 | 300.0 | 75.0 | 300.0 | 1000 | 0.12 | 62.0 |" in rows[0]
    assert "conv_slab_kernel<32, 32, true>" in rows[1] and rows[1].rstrip().endswith("| — | — | — | — |")


def test_winograd_restatement_equals_the_direct_conv():
    """tools/winograd_error.py: the float32 Winograd F(2,3) form on the dilation lattice (tap groups accumulated in the transform domain,
    the algorithm of csrc/conv_c64w.hip) reproduces the direct dilated conv for every (k, dilation) of the vocoder, odd lengths included,
    and the pmc_traffic class split of the one kernel template by its channel argument."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import winograd_error as W
    torch.manual_seed(1)
    for k in (3, 7, 11):
        for dil in (1, 3, 5):
            x = torch.randn(8, 203)
            w = torch.randn(8, 8, k) * (8 * k) ** -0.5
            ref = W.direct(x, w, dil, torch.float64)
            got = W.winograd23(x, w, dil)
            assert got.shape == ref.shape
            assert W.rel(got, ref) < 1e-6, (k, dil, W.rel(got, ref))
            assert W.rel(W.direct(x, w, dil, torch.float32), ref) < 1e-6
    import re
    name = "void ss::conv_c64w_kernel<true, 3, 128>(ss::GemmArgs, int, int)"
    assert re.search(r"conv_c64w_kernel<(true|false), \d+, (\d+)>", name).group(2) == "128"
