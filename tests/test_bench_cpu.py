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
