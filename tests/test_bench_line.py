"""bench.py's last stdout line is what the driver parses: it must stay small (round 5's 20 KB line was cut by the driver's capture and
recorded as unparsed) and carry the contract's keys, roofline and cpu_baseline.  CPU-only: the line is built from a canned result."""
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _canned():
    """The round-5 line (every leg, 20 KB) as the worst case; extended with the keys this round adds."""
    with open(os.path.join(ROOT, "profiles", "r05_bench.json.log")) as f:
        full = json.loads(f.read().strip().splitlines()[-1])
    full["multi_gpu"] = dict(rank_ms_per_step=[3.9] * 8, allreduce_ms=[0.11] * 8, allreduce_bytes=5616972, param_checksum=[1.0] * 8,
                             params_identical=True, exchange="async (side stream under the split tail)", allreduce_note="x" * 500, preflight={"a": 1})
    full["rccl_selftest"] = dict(allreduce_us=41.0, note="y" * 900)
    full["dropin"].update(max_over_median=1.4)
    full["bench_seconds"] = 58.0
    return full


def test_compact_line_is_small_and_complete():
    full = _canned()
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT, len(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["dtype"] == "f32"
    r = line["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches_per_step", "share_of_step"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind"):
        assert c[k] is not None, k
    assert "workload" in line["config"] and "model" not in line["config"]
    for k in ("f32_mfma_ms_per_step", "sa1_stage_frac_best", "config3_frac_hbm_path_bytes", "eval_ms_per_batch", "rccl_allreduce_us"):
        assert isinstance(line[k], float), k
    # no prose: every string value in the record is a short name
    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, (list, tuple)):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(x) for x in strings(line)) <= 160


def test_emit_prints_the_compact_line_last_and_alone_on_stdout(tmp_path):
    full = _canned()
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit(full, str(tmp_path / "extras.json"))
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) < bench.LINE_LIMIT
    rec = json.loads(lines[0])
    assert rec["roofline"]["frac"] == full["roofline"]["frac"] and rec["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
    extra_keys = [next(iter(json.loads(ln[len("extra "):]))) for ln in err.getvalue().splitlines() if ln.startswith("extra ")]
    assert "kernels" in extra_keys and "stages" in extra_keys and "roofline" in extra_keys
    saved = json.load(open(tmp_path / "extras.json"))
    assert saved["stages"] == full["stages"]


def test_compact_line_without_extras_or_baselines():
    line = bench.compact_line(dict(metric="m", value=1.0, unit="points/s", n_gpus=8, steps=2, warmup=1, ms_per_step=1.0, higher_is_better=True,
                                   scaling="weak", vs_baseline=None, dtype="f32", data="synthetic", config=dict(workload="w"), roofline=None,
                                   cpu_baseline=None))
    assert line["roofline"] is None and line["cpu_baseline"] is None and len(json.dumps(line)) < 1024
