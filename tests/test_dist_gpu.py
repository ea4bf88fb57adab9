"""Multi-GPU path on real devices: two ranks over `nccl` (= RCCL over xGMI on ROCm), one process per GPU, shard a
batch, round their shard with no communication and gather the packed cores once (plain and pipelined form); the
result equals the single-process rounding bit for bit.  Needs two visible GPUs (skipped on the 1-GPU test box), plus
`bench.py` under `torch.distributed.run --nproc-per-node 1` (the driver's launch form) on one GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import tntorch_amd as tn
    from tntorch_amd.dist_batch import gather_batch, round_tt_sharded, shard_range

    torch.manual_seed(0)  # every rank builds the same full batch (on the CPU), then keeps its block
    full = tn.randn([total, 8, 8, 8, 8], ranks_tt=6, batch=True, dtype=torch.float32)
    lo, hi = shard_range(total, world, rank)
    mine = [c[lo:hi].to(dev) for c in full.cores]
    out = round_tt_sharded(mine, rmax=3, algorithm="svd")
    t = tn.Tensor([c.clone() for c in mine], batch=True)
    t.round_tt(rmax=3, algorithm="svd")
    sizes = [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]
    h = gather_batch(t, dst=0, sizes=sizes, async_op=True)  # the pipelined form bench.py uses
    parts = h.wait()
    torch.cuda.synchronize()
    if rank == 0:
        assert len(parts) == world and [p.cores[0].shape[0] for p in parts] == sizes
        merged = [torch.cat([p.cores[k] for p in parts]) for k in range(len(out.cores))]
        assert all(torch.equal(a, b) for a, b in zip(merged, out.cores))
        q.put((dist.get_world_size(), [c.cpu().numpy() for c in out.cores]))
    else:
        assert out is None and parts is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("total", [300, 7])  # two-stream shards (>= 128 per rank) and a small ragged split
def test_nccl_world2_matches_single_process(total):
    import torch.multiprocessing as mp

    import tntorch_amd as tn

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    nranks, got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert nranks == 2
    torch.manual_seed(0)
    full = tn.randn([total, 8, 8, 8, 8], ranks_tt=6, batch=True, dtype=torch.float32)
    ref = tn.Tensor([c.cuda() for c in full.cores], batch=True)
    ref.round_tt(rmax=3, algorithm="svd")
    for a, b in zip(got, ref.cores):
        assert torch.equal(torch.from_numpy(a), b.cpu())  # sharding does not change a single bit


def test_bench_under_torchrun_single_rank():
    """The driver launches `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: with N = 1 this
    must produce the same kind of line as the plain invocation (small batch: this is a launch-path test, not a timing)."""
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "130", "--no-cpu-baseline", "--no-extras"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["nccl_ranks"] == 1 and d["unit"] == "cores/s" and d["value"] > 0
    assert d["parity"]["ok"], d["parity"]
    # at this small batch the latency-bound eigensolver can be the dominant kind: bench.py labels it "valu"
    assert d["roofline"]["bound"] in ("mfma", "hbm", "valu") and 0 < d["roofline"]["frac"] < 1


def _nccl1_worker(port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import tntorch_amd as tn
    from tntorch_amd.dist_batch import GatherSchedule, gather_batch

    torch.manual_seed(0)
    full = tn.randn([300, 8, 8, 8, 8], ranks_tt=6, batch=True, dtype=torch.float32)   # >= 128: two streams + result arena
    mine = [c.to(dev) for c in full.cores]
    outs = []
    for mode in ("step", "end", "none"):
        sched = GatherSchedule(mode, sizes=[300], dst=0, local_shortcut=False)
        last = None
        for k in range(3):
            t = tn.Tensor([c * (k + 1) for c in mine], batch=True)
            t.round_tt(rmax=3, algorithm="svd")
            sched.after_step(t)
            last = t
        parts = sched.drain()
        torch.cuda.synchronize()
        assert sched.gathers == {"step": 3, "end": 1, "none": 0}[mode]
        if mode == "none":
            assert parts is None
            continue
        assert len(parts) == 1 and parts[0].cores[0].shape[0] == 300
        # the receive buffer is NOT the send buffer: the data went through RCCL
        assert parts[0].cores[0].data_ptr() != last.cores[0].data_ptr()
        assert all(torch.equal(a, b) for a, b in zip(parts[0].cores, last.cores))
        outs.append(mode)
    h = gather_batch(last, dst=0, sizes=[300], async_op=True, local_shortcut=False)   # the pipelined form, directly
    p2 = h.wait()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(p2[0].cores, last.cores))
    q.put((dist.get_backend(), dist.get_world_size(), outs))
    dist.barrier()
    dist.destroy_process_group()


def test_nccl_world1_gather_through_the_collective():
    """One GPU, a real `nccl` (RCCL) process group of one rank: the packed view of the result arena, the receive buffers, the
    asynchronous `dist.gather`, its work handle and the three `GatherSchedule` policies of bench.py run through the collective
    (`local_shortcut=False`) instead of the single-process shortcut."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl1_worker, args=(_free_port(), q))
    p.start()
    backend, nranks, outs = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0 and backend == "nccl" and nranks == 1 and outs == ["step", "end"]


@pytest.mark.parametrize("mode", ["end", "step"])
def test_bench_forced_dist_single_rank(mode):
    """bench.py's N > 1 control flow (process group, gather policy through RCCL, barrier-fenced timing, max over ranks, the
    `gather` block with compute-only time) on ONE GPU: TTR_BENCH_FORCE_DIST=1 under torch.distributed.run with one rank."""
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TTR_BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
           "--batch", "130", "--no-cpu-baseline", "--no-extras", "--gather", mode]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    g = d["gather"]
    assert d["nccl_ranks"] == 1 and g["mode"] == mode and g["collectives_in_timed_region"] == (1 if mode == "end" else 3)
    assert g["alone_ms"] > 0 and g["compute_only_ms_per_step"] > 0 and d["parity"]["ok"]
