"""Host-side bookkeeping of bench.py (no GPU): which counter data of profiles/pmc_latest.json a bench line may quote."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_committed_counters_are_usable_or_declared_stale():
    """profiles/pmc_latest.json either describes the kernel sources in the tree (per kind) or bench.py says why `traffic` is null."""
    pmc, note = bench.load_pmc()
    if pmc is None:
        assert note
        return
    for k in bench.KIND_SOURCES:
        if k in pmc and k != "misc":
            assert pmc[k].get("hbm_bytes_per_step") is not None and "mfma_util" in pmc[k], k


def test_counters_are_kept_per_kernel_kind(tmp_path, capsys):
    now = bench.kind_shas()
    assert set(now) == set(bench.KIND_SOURCES) and now["qr_factor"] == now["qr_apply"] and now["eigh"] != now["qr_factor"]
    body = {k: {"hbm_bytes_per_step": 1.0, "mfma_util": 0.5} for k in bench.KIND_SOURCES}
    # (a) same build: everything is used
    f = tmp_path / "a.json"
    f.write_text(json.dumps({"source_sha": bench.source_sha(), "kind_sha": now, "_batch": 2048, **body}))
    pmc, note = bench.load_pmc(str(f))
    assert note is None and all(k in pmc for k in bench.KIND_SOURCES)
    # (b) another file changed since: only the kinds compiled from it (and the catch-all) are dropped
    then = dict(now, eigh="0" * 16, misc="1" * 16)
    f = tmp_path / "b.json"
    f.write_text(json.dumps({"source_sha": "f" * 16, "kind_sha": then, "_batch": 2048, **body}))
    pmc, note = bench.load_pmc(str(f))
    assert note is None and "eigh" not in pmc and "misc" not in pmc and "qr_factor" in pmc and "project" in pmc
    assert "eigh" in capsys.readouterr().err
    # (c) a file without per-kind hashes from another build, or with every kind stale: nothing is used
    f = tmp_path / "c.json"
    f.write_text(json.dumps({"source_sha": "f" * 16, "_batch": 2048, **body}))
    assert bench.load_pmc(str(f))[0] is None
    f = tmp_path / "d.json"
    f.write_text(json.dumps({"source_sha": "f" * 16, "kind_sha": {k: "0" * 16 for k in now}, "_batch": 2048, **body}))
    pmc, note = bench.load_pmc(str(f))
    assert pmc is None and "stale" in note
    assert bench.load_pmc(str(tmp_path / "missing.json"))[0] is None


def _stub_full_result():
    """A realistic full result object: round 4's own (20.6 KB, the one the driver could not parse), with every optional block."""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = json.load(open(os.path.join(here, "profiles", "r04_bench.json")))
    res["gather"] = {"mode": "end", "collectives_in_timed_region": 1, "bytes_per_peer_per_step": 3.26e9, "peers": 7,
                     "alone_ms": 42.123456, "compute_only_ms_per_step": 13.912345, "hidden_under_compute": None,
                     "root_inbound_GBs": 540.0, "per_link_GBs": 77.4, "root_receive_buffers_GB": 26.1,
                     "hbm_GB": {"total": 288.0, "free_now": 100.0, "reserved_by_torch": 60.0}}
    return res


def test_stdout_line_is_compact_and_carries_the_contract_fields(tmp_path, capsys, monkeypatch):
    """BENCH_r04.parsed was null: the line had grown to 20.6 KB.  The ONE stdout line stays below 4 KB whatever the full
    result holds, is valid JSON without NaN / Infinity, and keeps the fields the driver and the judge read."""
    res = _stub_full_result()
    assert len(json.dumps(res)) > 4 * bench.LINE_LIMIT           # (the stub really is the oversized object)
    line = bench.compact_line(res)
    assert len(line) < bench.LINE_LIMIT and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"] and "model" not in d["config"]
    assert d["dtype"] == "f32" and d["unit"] == "cores/s" and d["steps"] == res["steps"] and d["warmup"] == res["warmup"]
    assert abs(d["value"] - res["value"]) <= 1e-6 * res["value"] and abs(d["ms_per_step"] - res["ms_per_step"]) <= 1e-5 * res["ms_per_step"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert 0.0 < d["roofline"]["frac"] <= 1.0
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["parity"]["ok"] is True and set(d["configs"]) == {"c1", "c2", "c3", "c4"} and d["gather"]["alone_ms"] > 0
    # a pathological full result (huge blocks, NaN): optional blocks are dropped, the contract fields never
    res["extras"] = {f"x{i}": {"ms_per_step": float(i), "cores_per_s": 1.0} for i in range(400)}
    res["roofline"]["mfma_util"] = float("nan")
    line = bench.compact_line(res)
    d = json.loads(line)
    assert len(line) < bench.LINE_LIMIT and "extras" not in d and d["roofline"]["frac"] > 0 and "mfma_util" not in d["roofline"]
    # emit(): stdout carries exactly the compact line (last line), the full object goes to the sidecar and to stderr
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    bench.emit(_stub_full_result())
    cap = capsys.readouterr()
    out_lines = [x for x in cap.out.splitlines() if x.strip()]
    assert len(out_lines) == 1 and json.loads(out_lines[0])["roofline"]["frac"] > 0
    full = json.load(open(tmp_path / "profiles" / "bench_full_latest.json"))
    assert "roofline_per_kernel" in full and "configs" in full and "bench.py full result" in cap.err


def test_per_kind_roofline_is_on_executed_work():
    """`frac` of a kind = the larger of its two EXECUTED roofline fractions (census / measured time); the input-blind model stands
    beside it as `algorithmic` and may exceed 1 where launches skip work; kinds without a census carry no fraction."""
    B, steps = 2048, 5
    model = bench.kernel_model()
    prof = {k: {"ms": 10.0, "launches": 10} for k in ("qr_factor", "rotgram", "eigh", "misc")}
    prof["gemm"] = {"ms": 0.0, "launches": 0}
    work = {"qr_factor": {"flops": 0.3 * model["qr_factor"]["flops"] * B * steps, "bytes": 0.8 * model["qr_factor"]["bytes"] * B * steps},
            "rotgram": {"flops": 1e9, "bytes": 1e9}, "eigh": {"flops": 0.0, "bytes": 0.0}, "misc": {"flops": 0.0, "bytes": 0.0}}
    pk = bench.per_kind_roofline(prof, work, B, steps)
    assert set(pk) == {"qr_factor", "rotgram", "eigh", "misc"}
    q = pk["qr_factor"]
    assert abs(q["executed_share_of_algorithmic_flops"] - 0.3) < 1e-12 and q["frac"] == max(q["executed"]["mfma_frac"], q["executed"]["hbm_frac"])
    assert q["bound"] == ("mfma" if q["executed"]["mfma_frac"] >= q["executed"]["hbm_frac"] else "hbm")
    assert pk["rotgram"]["algorithmic"]["mfma_frac"] > pk["rotgram"]["executed"]["mfma_frac"]     # (the model prices skipped work)
    assert pk["eigh"]["frac"] is None and pk["eigh"]["bound"] == "valu" and pk["misc"]["frac"] is None
    hl = bench.headline_roofline(pk, "qr_factor", prof, steps)
    assert hl["input_aware"] and hl["frac"] == q["frac"] and hl["frac_algorithmic"] is not None and hl["traffic"] is None
    sw = bench.sweep_executed(pk, 10.0)
    assert sw["kinds"] == ["qr_factor", "rotgram"] and sw["flops_per_step"] > 0
    assert bench.headline_roofline(pk, "eigh", prof, steps)["frac"] is None
