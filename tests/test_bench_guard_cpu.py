"""bench.py's line guard without a GPU: the ONE JSON line reaches stdout exactly once whether the worker finishes, stalls in a
secondary workload past the deadline, or dies after the headline; nothing is printed if no headline was ever measured."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


import pytest  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_lines  # noqa: E402


@pytest.fixture(autouse=True)
def _extras_file_in_tmp(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "EXTRAS_FILE", str(tmp_path / "bench_extras.json"))


def _lines(capsys):
    """the headline lines on stdout (the full record travels on an earlier, prefixed line: bench_lines)"""
    out = capsys.readouterr().out
    if bench_lines.headline_lines(out):
        bench_lines.split(out)           # the contract: one headline, last, below 4 KB, the full record before it
    return bench_lines.headline_lines(out)


def test_worker_finishes_last_emitted_line_wins(capsys):
    def worker(emit):
        emit({"value": 1})
        emit({"value": 1, "extras": {"q3": 2}})
    assert bench.run_guarded(worker, deadline_s=30) == 0
    out = _lines(capsys)
    assert len(out) == 1 and json.loads(out[0])["value"] == 1 and "extras" not in json.loads(out[0])
    assert json.load(open(bench.EXTRAS_FILE)) == {"value": 1, "extras": {"q3": 2}}


def test_stalled_secondary_workload_is_cut_at_the_deadline(capsys):
    def worker(emit):
        emit({"value": 7})
        time.sleep(60)            # a secondary workload that never comes back
        emit({"value": -1})
    t0 = time.monotonic()
    assert bench.run_guarded(worker, deadline_s=1.0, poll_s=0.05) == 0
    assert time.monotonic() - t0 < 10
    out = _lines(capsys)
    assert len(out) == 1
    d = json.loads(out[0])
    assert d["value"] == 7 and "deadline" in d["note"]


def test_no_deadline_cut_before_a_headline_exists(capsys):
    def worker(emit):
        time.sleep(1.0)           # slow headline: the guard must keep waiting past the deadline
        emit({"value": 3})
    assert bench.run_guarded(worker, deadline_s=0.2, poll_s=0.05) == 0
    out = _lines(capsys)
    assert len(out) == 1 and json.loads(out[0])["value"] == 3


def test_no_cut_while_mandatory_parts_are_missing(capsys):
    def worker(emit):
        emit({"value": 9}, False)            # headline measured, CPU baseline still running when the deadline passes
        time.sleep(1.0)
        emit({"value": 9, "cpu_baseline": {"value": 1}}, True)
        time.sleep(60)                       # a stalled secondary workload
    t0 = time.monotonic()
    assert bench.run_guarded(worker, deadline_s=0.3, poll_s=0.05) == 0
    assert 0.9 < time.monotonic() - t0 < 10
    out = _lines(capsys)
    assert len(out) == 1 and json.loads(out[0])["cpu_baseline"] == {"value": 1}


def test_worker_dying_after_the_headline_still_reports_it(capsys):
    def worker(emit):
        emit({"value": 5})
        raise RuntimeError("secondary workload crashed")
    assert bench.run_guarded(worker, deadline_s=30) == 0
    out = _lines(capsys)
    assert len(out) == 1 and json.loads(out[0])["value"] == 5


def test_worker_dying_before_the_headline_prints_nothing_and_fails(capsys):
    def worker(emit):
        raise RuntimeError("no GPU")
    assert bench.run_guarded(worker, deadline_s=30) != 0
    assert _lines(capsys) == []


def test_pyarrow_yardstick_computes_q1(orc):
    """The third-party CPU yardstick of bench.py's cpu_baseline evaluates the same query as the oracle."""
    import numpy as np
    from polars_amd import datagen
    cols = datagen.lineitem_host(200_000, seed=4)
    cutoff = datagen.us(1998, 9, 2)
    t = bench.pyarrow_q1({k: cols[k] for k in datagen.LINEITEM_Q1_COLS}, cutoff).to_pydict()
    want = orc.q1(cols, cutoff)
    order = sorted(range(len(t["l_returnflag"])), key=lambda i: (t["l_returnflag"][i], t["l_linestatus"][i]))
    assert [t["l_returnflag"][i] for i in order] == want["l_returnflag"].tolist()
    assert [t["count_all"][i] for i in order] == want["count_order"].tolist() and [t["l_quantity_sum"][i] for i in order] == want["sum_qty"].tolist()
    assert np.allclose([t["charge_sum"][i] for i in order], want["sum_charge"], rtol=1e-9)
    assert np.allclose([t["l_discount_mean"][i] for i in order], want["avg_disc"], rtol=1e-9)


def test_failed_verification_makes_the_exit_code_non_zero(capsys):
    """A parity regression at full size must not ship with a green rc (round-2 review, Weak 2): the line is still printed."""
    def worker(emit):
        emit({"value": 1, "verified": {"ok": True}, "extras": {"tpch_q3_sf100": {"verified": {"ok": False, "max_rel_err": 1.0}}}})
    assert bench.run_guarded(worker, deadline_s=30) == bench.EXIT_PARITY
    assert len(_lines(capsys)) == 1
    assert bench.failed_verifications({"verified": {"ok": False}, "config": {"workload": "q1"}}) == ["q1"]
    assert bench.failed_verifications({"verified": {"ok": None}, "extras": {"a": {"verified": {"ok": True}}, "scan": {"files": {"snappy": {"verified": False}}}}}) == ["scan.snappy"]
    assert bench.failed_verifications({"extras": {"x": {"error": "boom"}}}) == []


def test_headline_line_stays_below_4k_on_the_largest_record_we_have():
    """Round 4's default run printed ONE 25 KB line and the driver, which parses the last line out of an 8 KB tail, recorded `parsed: null`.  The same record
    through the splitter: a headline below 4 KB that still carries metric / value / config / roofline / cpu_baseline / verified and one row per extra."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = json.load(open(os.path.join(root, "profiles", "r04", "bench_default_run.json")))
    assert len(json.dumps(full)) > 20000
    head = bench.headline_line(full, os.path.join(root, "bench_extras.json"))
    text = json.dumps(head)
    assert len(text.encode()) < bench_lines.HEADLINE_LIMIT, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "verified"):
        assert k in head, k
    assert head["value"] == full["value"] and head["ms_per_step"] == full["ms_per_step"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert head["roofline"][k] == full["roofline"][k], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in head["cpu_baseline"], k
    assert head["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and len(head["cpu_baseline"]["sample"]) <= 160
    assert head["config"]["workload"] == "tpch_q1_sf100" and head["extras_file"] == "bench_extras.json"
    assert set(head["extras_summary"]) == set(full["extras"])
    assert head["extras_summary"]["cfg5_dict_string_keys_1e9"]["frac"] == full["extras"]["cfg5_dict_string_keys_1e9"]["roofline"]["frac"]
    # a pathological record (every string huge) still comes out below the limit
    fat = dict(full, config={k: "x" * 5000 for k in "abcdef"}, note="y" * 10000)
    assert len(json.dumps(bench.headline_line(fat, None)).encode()) < bench_lines.HEADLINE_LIMIT
