"""The bench line is a record the driver must be able to read: one stdout line, strict JSON, well under 8 KB, carrying the contract's
keys with `roofline` and `cpu_baseline` (r04's 28.8 KB line came back from the driver unparsed).  Built here from a canned full
result (the committed profiles/r04_bench.json) - no GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

SCHEMA = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
          "roofline", "cpu_baseline")


def _strict(text):
    def bad(c):
        raise ValueError("non-finite constant %s in the line" % c)
    return json.loads(text, parse_constant=bad)


def _canned():
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    full["step_roofline"] = {"bound": "hbm", "achieved": 805.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1006, "algorithmic_bytes_per_step": 15.1e9, "traffic": None, "note": "x" * 400}
    return full


def test_line_is_small_strict_json_with_the_contract_keys():
    text = bench.compact_line(_canned())
    assert "\n" not in text
    assert len(text.encode()) < 8192, len(text)
    line = _strict(text)
    for k in SCHEMA:
        assert k in line, k
    assert line["value"] > 0 and line["unit"] == "frames/s" and line["higher_is_better"] is True
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-4
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["step_roofline"]["frac"] > 0
    assert line["metric_ba"]["roofline"]["bound"] == "mfma" and "cpu_baseline" in line["metric_ba"]
    # numeric-only per-stage table: no prose
    for row in line["rooflines"]:
        assert len(row) == len(line["rooflines_columns"])
        assert all(v is None or isinstance(v, (int, float)) for v in row[2:])
    assert max(len(v) for v in _strings(line)) <= 200


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, list):
        for v in o:
            yield from _strings(v)


def test_line_survives_nan_and_oversized_input():
    full = _canned()
    full["value"] = float("nan")
    full["rooflines"] = full["rooflines"] * 60            # a table that would blow the limit is dropped, the schema keys stay
    text = bench.compact_line(full)
    assert len(text.encode()) < 8192
    line = _strict(text)
    assert line["value"] is None and "roofline" in line and "cpu_baseline" in line


def test_emit_prints_the_line_last(tmp_path, capsys, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(_canned())
    cap = capsys.readouterr()
    lines = cap.out.strip().splitlines()
    assert len(lines) == 1 and _strict(lines[0])["full"] == "bench_full.json"
    side = _strict(open(tmp_path / "bench_full.json").read())
    assert "secondary_metrics" in side and "rooflines" in side


def test_rendered_sequences_are_cached_and_read_back(tmp_path, monkeypatch):
    """bench.make_sequences: a rendered sequence is written once (atomically) under PS_SEQ_CACHE and the second request reads it in this process -
    what lets a run under the counter tool (no child processes) and the other ranks skip the rendering."""
    import numpy as np
    monkeypatch.setenv("PS_SEQ_CACHE", str(tmp_path))
    job = ("lateral", 2, 41, 0, "synthetic", 2)
    assert bench._load_cached(job) is None
    a = bench._make_one(job)
    assert len(list(tmp_path.iterdir())) == 1 and not any(p.name.endswith(".tmp") for p in tmp_path.iterdir())
    b = bench._load_cached(job)
    assert b is not None and np.array_equal(a["left"], b["left"]) and np.array_equal(a["masks"], b["masks"])
    assert bench._seq_cache_path(("lateral", 2, 42, 0, "synthetic", 2)) != bench._seq_cache_path(job)


def test_r06_line_carries_the_boundary_numbers_and_a_fraction_for_every_stage():
    """VERDICT r05 items 1 and 4, on the committed full result of the round (profiles/r06_bench_full.json): the compact line holds the one-sequence-
    per-GPU number of BASELINE configs[4] and the drop-in boundary's per-frame time with its split, `traffic` says where it comes from, and no stage
    that is 1 % of the step or more is left without a roofline fraction."""
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_full.json")))
    text = bench.compact_line(full)
    assert len(text.encode()) < 8192, len(text)
    line = _strict(text)
    c5 = line["config5_one_sequence_per_gpu"]
    assert 0.3 < c5["ms_per_frame"] < 5 and c5["camera_chain_only_ms_per_frame"] < c5["ms_per_frame"] and c5["tracked"] == 154
    di = line["drop_in_api"]
    sp = di["cpp_shim_camera_chain"]["split_ms"]
    parts = sp["extract_call_ms"] + sp["stereo_call_ms"] + sp["search_call_ms"] + sp["pose_call_ms"] + sp["host_marshalling_ms"]
    assert abs(parts - sp["wall_ms"]) < 0.02 * sp["wall_ms"]                       # the split adds up to the frame
    assert sp["extract_kernels_ms"] < sp["extract_call_ms"] and sp["pose_kernels_ms"] < sp["pose_call_ms"]
    assert di["cpp_shim_camera_chain"]["frames_with_pose"] == 154 and di["per_call_slot_chain"]["c_abi_calls_per_frame"] >= 10
    assert line["roofline"]["traffic_source"].startswith("profiles/")
    step = sum(r[2] for r in line["rooflines"])
    for stage, bound, ms, frac, _ in line["rooflines"]:
        if ms >= 0.01 * step:
            assert frac is not None and 0 < frac <= 1.0, (stage, bound, ms, frac)
    assert all(v is not None for v in line["kernels_alone"]["frac"].values())
