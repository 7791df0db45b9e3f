"""The driver parses the LAST stdout line of bench.py; round 5's 21.5 KB line was dropped (BENCH_r05.json: parsed null).

bench.compact() is what is printed there now: the contract's keys + scalar config keys + roofline + cpu_baseline."""
import glob
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

FULL = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_default.json")))


def _load(p):
    with open(p) as f:
        txt = f.read().strip().splitlines()[-1]
    return json.loads(txt)


@pytest.mark.parametrize("path", FULL, ids=[os.path.basename(p) for p in FULL])
def test_compact_line_of_recorded_runs(path):
    res = _load(path)
    if "roofline" not in res:
        pytest.skip("an early record without roofline")
    line = bench.compact(res)
    assert "\n" not in line
    assert len(line) < bench.COMPACT_LINE_LIMIT
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data"):
        assert k in out, k
    assert out["value"] == pytest.approx(res["value"], rel=1e-4)
    assert out["ms_per_step"] == pytest.approx(res["ms_per_step"], rel=1e-4)
    assert isinstance(out["config"]["workload"], str) and len(out["config"]["workload"]) <= 120
    rf = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3)
    if "cpu_baseline" in res:
        cb = out["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in cb, k
        assert len(cb["sample"]) <= 160
    # nothing nested beyond config / roofline / cpu_baseline
    for k, v in out.items():
        if k not in ("config", "roofline", "cpu_baseline", "collectives", "work_queue_check"):
            assert not isinstance(v, (dict, list)), k
    for v in out["config"].values():
        assert not isinstance(v, dict)


def test_compact_line_survives_a_bloated_config():
    res = _load(FULL[-1])
    for i in range(400):
        res["config"]["extra_key_%03d" % i] = 1.0 / (i + 3)
    line = bench.compact(res)
    assert len(line) <= bench.COMPACT_LINE_LIMIT
    out = json.loads(line)
    assert "workload" in out["config"] and "roofline" in out and "cpu_baseline" in out


def test_emit_prints_compact_last_and_writes_full(tmp_path, capsys):
    res = _load(FULL[-1])
    p = str(tmp_path / "sub" / "full.json")
    bench.emit(res, p)
    cap = capsys.readouterr()
    last = cap.out.strip().splitlines()[-1]
    out = json.loads(last)
    assert len(last) < bench.COMPACT_LINE_LIMIT
    assert out["full_record"] == p
    with open(p) as f:
        full = json.loads(f.read())
    assert "configs" in full and "c2_runs" in full            # the nested sub-records live in the full record
    assert "configs" not in out and "e2e_bigwig_genome" not in out
    assert cap.err.startswith("bench full record: ")


def test_nan_and_inf_do_not_break_the_line():
    res = _load(FULL[-1])
    res["auc_check"] = float("nan")
    res["config"]["c3_hbm_frac"] = float("inf")
    out = json.loads(bench.compact(res))            # strict JSON: no NaN / Infinity tokens
    assert out["auc_check"] is None and out["config"]["c3_hbm_frac"] is None
    assert "NaN" not in bench.compact(res) and "Infinity" not in bench.compact(res)
