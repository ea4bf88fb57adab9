#!/usr/bin/env python3
"""Round-2 probe (run on the GPU box through gpurun): instruction-rate microbenchmarks, A/B timing of the QR panel
variants on the metric's level-0 shape, and the whole round_tt step under every variant.  Prints JSON lines."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def microbench():
    exe = "/tmp/ttr_microbench"
    src = os.path.join(ROOT, "tools", "microbench.hip")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-Wno-unused-value", src, "-o", exe], check=True)
    print(subprocess.run([exe], capture_output=True, text=True, timeout=300).stdout, flush=True)


def ev_time(fn, reps):
    import torch
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def qr_variants(B):
    import torch
    from tntorch_amd import _hip as h
    g = torch.Generator(device="cuda").manual_seed(1)
    Rm = torch.randn(B, 64, 64, generator=g, device="cuda").triu()
    core = torch.randn(B, 64, 64, 64, generator=g, device="cuda")
    P = (Rm[:2].double() @ core[:2].double().reshape(2, 64, -1)).reshape(2, 4096, 64)
    stamps = torch.zeros(64, dtype=torch.int64, device="cuda")
    for v in (0, 1):
        h.set_knob(h.KNOB_QR_PANEL, v)
        try:
            ms = ev_time(lambda: h.qr_factor_pushed(Rm, core), 5)
            h.prof_enable(True)
            f = h.qr_factor_pushed(Rm, core)
            torch.cuda.synchronize()
            prof = h.prof_collect()
            h.prof_enable(False)
            h.lib().ttr_debug_set_qr_stamps(stamps.data_ptr())
            stamps.zero_()
            f = h.qr_factor_pushed(Rm, core)
            torch.cuda.synchronize()
            h.lib().ttr_debug_set_qr_stamps(None)
            st = stamps.cpu().tolist()
            st = [b - st[0] for b in st if b > 0]
            Q = h.qr_apply(f)[:2].double()
            R = f.R[:2].double()
            orth = (Q.transpose(1, 2) @ Q - torch.eye(64, dtype=torch.float64, device="cuda")).abs().max().item()
            rec = ((Q @ R - P.cuda()).abs().max() / P.abs().max()).item()
            C32 = f.R[:, :, :32].contiguous()
            ms_apply = ev_time(lambda: h.qr_apply(f, C32), 3)
            print(json.dumps({"qr_variant": v, "B": B, "factor_ms_all_levels": ms, "factor_launch_ms": prof["qr_factor"],
                              "orth": orth, "recon": rec, "stamps_block0": st[:40], "apply32_ms": ms_apply}), flush=True)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"qr_variant": v, "error": repr(e)}), flush=True)
        finally:
            h.set_knob(h.KNOB_QR_PANEL, 1)


def step_variants(B, steps=3):
    import torch
    import tntorch_amd as tn
    from tntorch_amd import _hip as h
    sys.path.insert(0, ROOT)
    import bench
    inp = bench.make_input(B, torch.device("cuda"), seed=1234)
    for v in (0, 1):
        h.set_knob(h.KNOB_QR_PANEL, v)
        try:
            def step():
                t = tn.Tensor(inp, batch=True)
                t.round_tt(rmax=32)
                return t
            step(); step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            from tntorch_amd import _hipops
            _hipops.STREAM_CHUNKS_ENABLED = False
            h.prof_enable(True)
            step()
            torch.cuda.synchronize()
            prof = h.prof_collect()
            h.prof_enable(False)
            _hipops.STREAM_CHUNKS_ENABLED = True
            print(json.dumps({"step_variant": v, "B": B, "ms_per_step": ms, "cores_per_s": B * 8 / ms * 1e3,
                              "kernel_ms": {k: round(x["ms"], 3) for k, x in prof.items()},
                              "launches": {k: x["launches"] for k, x in prof.items()}}), flush=True)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"step_variant": v, "error": repr(e)}), flush=True)
        finally:
            h.set_knob(h.KNOB_QR_PANEL, 1)


def sweep_kernels(B):
    """The R2L kernels on the metric's bond shape (64 x 2048 per item): fused kernels vs the generic GEMM they replace."""
    import torch
    from tntorch_amd import _hip as h
    g = torch.Generator(device="cuda").manual_seed(2)
    M = torch.randn(B, 64, 2048, generator=g, device="cuda")
    V1 = torch.linalg.qr(torch.randn(B, 64, 64, generator=g, device="cuda"))[0].contiguous()
    sig = torch.rand(B, 64, generator=g, device="cuda") + 0.5
    res = {"B": B}
    res["gemm_gram_ms"] = ev_time(lambda: h.gemm(M, M, transB=True), 5)
    res["rowgram_ms"] = ev_time(lambda: h.rowgram(M), 5)
    res["gemm_rotate_plus_gram_ms"] = ev_time(lambda: h.gemm(h.gemm(V1, M, transA=True), h.gemm(V1, M, transA=True), transB=True), 3)
    res["rotgram_ms"] = ev_time(lambda: h.rowgram(M, V1), 5)
    res["gemm_project_ms"] = ev_time(lambda: h.gemm(V1[:, :, :32], M, transA=True, rowscale=sig, rowscale_mode=h.SCALE_DIV), 5)
    res["project_ms"] = ev_time(lambda: h.project(M, V1, V1, sig, 32, True), 5)
    res["eigh_tridiag_ms"] = None
    G = h.gemm(M, M, transB=True)
    res["eigh_tridiag_ms"] = ev_time(lambda: h.eigh_trunc(G, h.EIG_RAW, False, 0.0, 64, abs_floor=h.SOLVER_TRIDIAG), 3)
    Vt, _, _ = h.eigh_trunc(G, h.EIG_RAW, False, 0.0, 64, abs_floor=h.SOLVER_TRIDIAG)
    G2 = h.rowgram(M, Vt)
    res["eigh_live_ms"] = ev_time(lambda: h.eigh_trunc(G2, h.EIG_RAW, False, 0.0, 32, abs_floor=h.SOLVER_JACOBI_LIVE), 3)
    res["bytes_M"] = M.numel() * 4
    print(json.dumps({"sweep_kernels": res}), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["micro", "qr", "sweep", "step"]
    if "micro" in what:
        try:
            microbench()
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"microbench_error": repr(e)}), flush=True)
    if "qr" in what:
        qr_variants(2048)
    if "sweep" in what:
        try:
            sweep_kernels(2048)
            sweep_kernels(64)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"sweep_error": repr(e)}), flush=True)
    if "step" in what:
        step_variants(2048)
        step_variants(64)
        step_variants(1, steps=10)
