#!/usr/bin/env python3
"""Benchmark of the TT rounding hot path on MI355X.

Workload (BASELINE.json `metric`, SURVEY 8d "M"): round_tt(rmax=32) of 64^8 tensors held in TT
format with rank 64 (t = g+g, g = randn-core TT of rank 32, float32) -- a dense 64^8 tensor
(1.1 PB) cannot exist, so "64^8 -> rank 32" is a TT-to-TT rounding.  One "step" rounds a batch
of B independent tensors resident in HBM (B per GPU; weak scaling over ranks) and, for N > 1,
gathers the rounded cores on rank 0 with a single RCCL gather.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the definition of every field).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CORES, MODE, R_IN, R_OUT = 8, 64, 64, 32
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FLOP_PER_TENSOR = 8.72e8   # SURVEY 8d, algorithmic flops of one 64^8 r64->r32 rounding
BYTES_PER_TENSOR = 2.69e7  # SURVEY 8d, algorithmic bytes (every core read once / written once per sweep)


def make_input(B, device, seed):
    """t = g + g with g = tn.randn([64]*8, ranks_tt=32) per batch item, built on the device."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    r = [1] + [R_OUT] * (N_CORES - 1) + [1]
    cores = []
    for k in range(N_CORES):
        g = torch.randn((B, r[k], MODE, r[k + 1]), generator=gen, device=device, dtype=torch.float32)
        if k == 0:
            c = torch.cat([g, g], dim=-1)
        elif k == N_CORES - 1:
            c = torch.cat([g, g], dim=-3)
        else:
            z = torch.zeros_like(g)
            c = torch.cat([torch.cat([g, z], dim=-1), torch.cat([z, g], dim=-1)], dim=-3)
        cores.append(c.contiguous())
    return cores


def algorithmic_bytes_per_tensor():
    """Bytes each kernel kind must move per tensor GIVEN ITS INTERFACE (float32, M workload): operands read
    once, results written once.  (SURVEY 8d's 2.69e7 B/tensor is the fully fused lower bound for the whole
    sweep and is reported separately as `whole_sweep_hbm_frac`.)"""
    s = 4
    mid = R_IN * MODE * R_IN
    first = 1 * MODE * R_IN
    last = R_IN * MODE * 1
    qr_factor = s * (first + (N_CORES - 2) * mid)                     # every core but the last enters one QR
    qr_apply = s * (1 * MODE * R_OUT + (N_CORES - 2) * R_IN * MODE * R_OUT)  # Q [U sigma; 0]: (R I) x 32 per core
    push = s * (2 * (N_CORES - 2) * mid + 2 * last + (N_CORES - 1) * R_IN * R_IN)    # R @ next core: read + write
    n_big = MODE * R_OUT                                              # right unfolding of a middle core: 64 x 2048
    per_bond = s * (R_IN * n_big                                      # Gram(M)
                    + 2 * R_IN * n_big                                # V1^T M: read + write
                    + R_IN * n_big                                    # Gram(M1)
                    + R_IN * n_big + R_OUT * n_big)                   # projection: read M1, write V^T
    last_bond = s * (4 * R_IN * MODE + R_IN * MODE + R_OUT * MODE)
    gemm = push + (N_CORES - 2) * per_bond + last_bond
    eigh = 2 * (N_CORES - 1) * s * 2 * R_IN * R_IN                    # two Jacobi passes per bond: G in, V out
    return {"qr_factor": qr_factor, "qr_apply": qr_apply, "gemm": gemm, "eigh": eigh}


def cpu_baseline(budget_s=25.0):
    """Time the CPU oracle (restatement of the reference, same LAPACK calls) on a bounded sample.

    The reference's small-matrix LAPACK calls do not scale with threads (128 MKL threads are ~40x SLOWER
    than 8 on this workload), so a few thread counts are tried and the FASTEST is reported: the baseline
    is the best the CPU path can do on this box, not a strawman.
    """
    import oracle

    torch.manual_seed(0)
    g = oracle.tt_randn([MODE] * N_CORES, R_OUT, dtype=torch.float32)
    inp = oracle.tt_add(g, g)
    ncpu = os.cpu_count() or 8
    cands = sorted({t for t in (4, 8, 16, 32) if t <= ncpu})  # >32 MKL threads is minutes per tensor on this workload
    saved = torch.get_num_threads()
    t_start = time.perf_counter()
    best = {}
    trials = {}
    for alg in ("eig", "svd"):
        for nt in cands:
            if time.perf_counter() - t_start > budget_s * (0.5 if alg == "eig" else 1.0):
                break
            torch.set_num_threads(nt)
            oracle.round_tt(inp, rmax=R_OUT, algorithm=alg)  # warm-up (MKL init / thread pool)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                oracle.round_tt(inp, rmax=R_OUT, algorithm=alg)
                ts.append(time.perf_counter() - t0)
                if ts[-1] > 1.5:
                    break
            med = sorted(ts)[len(ts) // 2]
            trials[f"{alg}@{nt}"] = med
            if alg not in best or med < best[alg][0]:
                best[alg] = (med, nt)
    torch.set_num_threads(saved)
    fast_alg = min(best, key=lambda a: best[a][0])
    sec, nt = best[fast_alg]
    return {
        "value": N_CORES / sec,
        "unit": "cores/s",
        "cores": nt,
        "kind": "port",
        "sample": f"oracle.round_tt(rmax=32) of ONE 64^8 rank-64 float32 TT (the metric's unit), median of <=3 runs per "
                  f"(algorithm, thread count) over threads {cands}; reported: fastest = algorithm='{fast_alg}' at {nt} threads",
        "algorithm": fast_alg,
        "sec_per_tensor": {a: best[a][0] for a in best},
        "threads": {a: best[a][1] for a in best},
        "trials_sec": trials,
        "gflops": FLOP_PER_TENSOR / sec / 1e9,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=2048,
                    help="tensors per GPU per step (2048 = eight single-wave 64x64 eigenproblems per CU; 13 GB of cores)")
    ap.add_argument("--algorithm", default="svd", choices=["svd", "eig"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-stream", action="store_true",
                    help="issue the timed region on one stream too (for rocprofv3 kernel traces: with sub-batch "
                         "streams the traced kernel durations overlap)")
    args = ap.parse_args()

    import torch.distributed as dist

    import tntorch_amd as tn
    from tntorch_amd import _hip
    from tntorch_amd.dist_batch import gather_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs a torch.distributed.run launch with WORLD_SIZE={args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    _hip.lib()  # fail loudly if the kernels are not built
    from tntorch_amd import _hipops
    if args.single_stream:
        _hipops.STREAM_CHUNKS_ENABLED = False

    B = args.batch
    inp = make_input(B, dev, seed=1234 + rank)

    sizes = [B] * world
    pending = [None]

    def step():
        """Round the local batch; for N > 1 hand the rounded cores to the (asynchronous) gather.  The gather
        of step k runs on RCCL's stream under the compute of step k+1; at most one gather is in flight."""
        t = tn.Tensor(inp, batch=True)
        t.round_tt(rmax=R_OUT, algorithm=args.algorithm)
        if world > 1:
            if pending[0] is not None:
                pending[0].wait()
            pending[0] = gather_batch(t, dst=0, sizes=sizes, async_op=True)
        return t

    def drain():
        if pending[0] is not None:
            res = pending[0].wait()
            pending[0] = None
            return res
        return None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    gathered = drain()  # every step's gather has completed inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        if rank == 0:  # the root really holds every rank's rounded cores
            assert gathered is not None and len(gathered) == world
            assert all(list(g.ranks_tt) == [1] + [R_OUT] * (N_CORES - 1) + [1] and g.cores[0].shape[0] == B for g in gathered)
            assert torch.isfinite(gathered[-1].cores[3]).all()

    # ---- per-kernel device time over an identical pass (HIP events on the launch stream).  The timed region
    # above runs sub-batches on several streams so that kernels overlap; here every kernel must run alone for
    # its duration to mean anything, so the same work is issued on ONE stream.
    _hipops.STREAM_CHUNKS_ENABLED = False
    _hip.prof_enable(True)
    for _ in range(args.steps):
        t = tn.Tensor(inp, batch=True)
        t.round_tt(rmax=R_OUT, algorithm=args.algorithm)
    torch.cuda.synchronize()
    prof = _hip.prof_collect()
    _hip.prof_enable(False)
    _hipops.STREAM_CHUNKS_ENABLED = not args.single_stream

    if rank == 0:
        if out is not None:
            assert list(out.ranks_tt) == [1] + [R_OUT] * (N_CORES - 1) + [1], out.ranks_tt
        tensors = B * world * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        cores_per_s = tensors * N_CORES / elapsed
        abytes = algorithmic_bytes_per_tensor()
        kinds = {k: v for k, v in prof.items() if v["launches"] > 0 and k in abytes}
        dom = max(kinds, key=lambda k: kinds[k]["ms"])
        launches = kinds[dom]["launches"]
        avg_launch_ms = kinds[dom]["ms"] / launches
        bytes_per_launch = abytes[dom] * B * args.steps / launches
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                traffic = pmc.get(dom, {}).get("hbm_bytes_per_launch")
                if traffic is not None and pmc.get("_batch"):
                    traffic = traffic * B / pmc["_batch"]
            except Exception:
                traffic = None
        per_kind = {
            k: {"ms_per_step": v["ms"] / args.steps,
                "achieved_GBs": abytes[k] * B * args.steps / (v["ms"] * 1e-3) / 1e9,
                "frac_of_hbm_peak": abytes[k] * B * args.steps / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
            for k, v in kinds.items()
        }
        res = {
            "metric": "TT rounding 64^8 rank-64 -> rank-32 (round_tt rmax=32), cores/s",
            "value": cores_per_s,
            "unit": "cores/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "round_tt(rmax=32) of 64^8 TT tensors, rank 64 (g+g, g randn rank 32), fp32, batch-resident in HBM",
                "tensors_per_gpu_per_step": B,
                "algorithm": args.algorithm,
                "streams_per_gpu": 1 if args.single_stream else 2,
                "parallelism": f"batch-sharded x{world}, one async RCCL gather of packed cores per step (overlapped with the next step)" if world > 1 else "single GPU",
            },
            "tensors_per_s": tensors / elapsed,
            "gflops": FLOP_PER_TENSOR * tensors / elapsed / 1e9,
            "whole_sweep_hbm_frac": BYTES_PER_TENSOR * tensors / elapsed / 1e9 / HBM_PEAK_GBS,
            "roofline": {
                "kernel": dom,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "avg_launch_ms": avg_launch_ms,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "launches": launches,
            },
            "roofline_per_kernel": per_kind,
            "kernel_ms_per_step": {k: v["ms"] / args.steps for k, v in prof.items() if v["launches"] > 0},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline()
            res["cpu_baseline"] = cb
            res["speedup_vs_cpu_best"] = cores_per_s / cb["value"]
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
