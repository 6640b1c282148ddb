"""The bench contract's ONE stdout line (the round-5 line was 21.4 KB and the driver could not parse it): whatever
the full result holds, `bench.compact_line` stays under LINE_BYTE_CAP (6 KB), is plain JSON, and carries `roofline` and
`cpu_baseline` as numbers; the full result goes to the sidecar file.  Fixture: the round-5 full result (profiles/r05_bench.json)."""
import copy
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _full():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))


def test_the_line_of_a_full_result_is_under_the_cap_and_keeps_the_contract():
    full = _full()
    assert len(json.dumps(full)) > 20000                       # the fixture IS the line the driver could not parse
    text = json.dumps(bench.compact_line(full))
    assert len(text) < bench.LINE_BYTE_CAP and "\n" not in text
    line = json.loads(text)
    assert all(k in line for k in CONTRACT) and "dropped" not in line
    r = line["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert all(isinstance(r[k], (int, float)) for k in ("achieved", "peak", "frac", "frac_issued", "launches", "avg_launch_us",
                                                         "algo_gflop_per_launch", "traffic", "traffic_over_algorithmic", "mfma_util_pct"))
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 16 and c["value"] == full["cpu_baseline"]["value"] and isinstance(c["sample"], str)
    assert c["oracle_check"]["ids_identical_to_float32_oracle"] is True and line["near_tie_rows"] == 0
    assert line["config"]["workload"] and "model" not in line["config"]
    assert line["pack_invariance"] == {"alone_equals_in_pack_bitwise": True}
    assert line["per_rank"] == [{"rank": 0, "wall_s": 1.97954, "audio_s": 12736.66, "utterances": 2560}]
    assert line["rccl"]["world"] == 1 and line["rccl"]["results_ok"] is True
    for k in ("streaming_320ms", "multilingual", "soak"):      # one number each
        assert isinstance(line[k], float)
    for k in ("dispatch", "process_census", "roofline_second_kernel", "multi_gpu_note", "host_placement"):
        assert k not in line                                   # sidecar material


def test_the_line_of_an_8_gpu_result_is_under_the_cap():
    full = _full()
    full["n_gpus"] = 8
    full["per_rank"] = [dict(full["per_rank"][0], rank=r) for r in range(8)]
    for k in ("cpu_baseline", "streaming_320ms", "multilingual", "soak", "bf16x3"):
        full[k] = None                                         # N > 1: no optional legs
    full["rccl"] = dict(full["rccl"], world=8)
    text = json.dumps(bench.compact_line(full))
    line = json.loads(text)
    assert len(text) < bench.LINE_BYTE_CAP and len(line["per_rank"]) == 8 and line["cpu_baseline"] is None and line["roofline"]["frac"] > 0


def test_a_bloated_result_drops_optional_keys_instead_of_passing_the_cap():
    full = copy.deepcopy(_full())
    full["n_gpus"] = 64
    full["per_rank"] = [dict(full["per_rank"][0], rank=r) for r in range(64)]
    full["rccl"] = {"error": "x" * 5000}
    text = json.dumps(bench.compact_line(full))
    line = json.loads(text)
    assert len(text) < bench.LINE_BYTE_CAP and "per_rank" in line["dropped"]
    assert all(k in line for k in CONTRACT) and len(line["rccl"]["error"]) <= 120


def test_failed_optional_legs_are_null_numbers_not_objects():
    full = _full()
    full["multilingual"] = {"value": None, "skipped": "does not fit"}
    full["soak"] = None
    line = bench.compact_line(full)
    assert line["multilingual"] is None and "soak" not in line


def test_gpus_8_dry_plan_prints_a_parseable_line_under_the_cap_and_a_sidecar(tmp_path):
    """The driver's SCALE command shape on CPU: stdout is exactly one JSON line < 6 KB, the full result (per-rank placement, the
    collectives' timings) is in the sidecar."""
    side = str(tmp_path / "detail.json")
    env = dict(os.environ, SS_BENCH_DETAIL=side, SS_BENCH_DRY_DETAIL="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-plan"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and len(lines[0]) < bench.LINE_BYTE_CAP
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["dry_plan"] and len(line["per_rank"]) == 8 and line["rccl"]["world"] == 8 and line["rccl"]["results_ok"]
    assert line["detail"] and "dropped" not in line
    full = json.load(open(side))
    assert "numa_node" in full["per_rank"][0] and full["comm"]["barrier_us"] > 0
    # a harness that merges the two streams still finds ONE candidate: no stderr line parses as an object with a "metric"
    for ln in r.stderr.splitlines():
        if ln.lstrip().startswith("{") or '{"metric"' in ln:
            try:
                assert "metric" not in json.loads(ln[ln.index("{"):]), ln[:200]
            except ValueError:
                pass


def test_the_streaming_mode_line_is_under_the_cap():
    """`bench.py --mode streaming` prints its own compact line (four agent configurations, CPU leg, long-prefix sweep)."""
    full = _full()["streaming_320ms"]
    full.update(metric="simultaneous S2ST fr-en", mode="streaming", higher_is_better=True, n_gpus=1, dtype="f32", data="synthetic")
    full["incremental_search_to_cap"] = full["incremental"]
    text = json.dumps(bench.compact_streaming_line(full, "bench_detail.json"))
    line = json.loads(text)
    assert len(text) < bench.LINE_BYTE_CAP and line["value"] == full["value"] and line["incremental"]["ms_per_read_call_mean"] > 0
    assert set(line["long_prefix_speedup_total"]) == {"15", "30"} and line["cpu_baseline"]["kind"] == "port"
