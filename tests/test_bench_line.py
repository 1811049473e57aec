"""The line bench.py prints for the driver: ONE short JSON line, the last of stdout (round 4's 24.9 KB line was not parsed).

CPU tier: compact_line() on full records of earlier runs (profiles/r0N_bench_default.json are what bench.py's `result` dict held)."""
import glob
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


RECORDS = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[4-9]_bench_default.json")) + glob.glob(os.path.join(ROOT, "profiles", "r0[5-9]_bench_full.json")))


@pytest.mark.parametrize("path", RECORDS, ids=[os.path.basename(p) for p in RECORDS])
def test_line_is_short_and_round_trips(path):
    b = _bench()
    full = json.load(open(path))
    line = b.compact_line(full)
    assert "\n" not in line
    assert len(line) < 8000 == b.LINE_LIMIT
    d = json.loads(line)
    # the contract's keys, unchanged in meaning
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["value"] == pytest.approx(full["value"], rel=1e-5)
    assert d["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert d["config"]["workload"] == full["config"]["workload"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert d["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["parity"]["bit_identical"] is full["parity"]["bit_identical"]
    for name, leg in full.get("secondary", {}).items():
        assert d["secondary"][name]["value"] == pytest.approx(leg["value"], rel=1e-5)
        assert d["secondary"][name]["frac"] == pytest.approx(leg["roofline"]["frac"], rel=1e-5)


def test_line_sheds_keys_rather_than_grow():
    """Whatever the legs hold (long error strings, many legs), the line stays under the limit and keeps the headline."""
    b = _bench()
    full = json.load(open(RECORDS[-1]))
    full["secondary"] = {("leg%02d" % i): dict(v, config=dict(v["config"], workload="w" * 500)) for i in range(24) for v in [full["secondary"]["c3"]]}
    full["cpu_baseline"]["sample"] = "s" * 5000
    line = b.compact_line(full)
    assert len(line) < b.LINE_LIMIT
    d = json.loads(line)
    assert d["roofline"]["frac"] and d["cpu_baseline"]["value"] and d["value"]


def test_non_finite_numbers_do_not_break_the_line():
    b = _bench()
    full = json.load(open(RECORDS[-1]))
    full["roofline"]["frac_necessary"] = float("nan")
    full["secondary"]["c3"]["value"] = float("inf")
    d = json.loads(b.compact_line(full))          # strict JSON: NaN / Infinity would not load in other parsers
    assert "NaN" not in b.compact_line(full) and "Infinity" not in b.compact_line(full)
    assert d["secondary"]["c3"]["value"] is None
