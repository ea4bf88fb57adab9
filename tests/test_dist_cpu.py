"""Multi-GPU path on CPU: world_size-2 gloo processes shard a batch, round their shard with no
communication and gather the packed cores once; the result equals the single-process rounding."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tntorch_amd as tn
    from tntorch_amd.dist_batch import round_tt_sharded, shard_range

    torch.manual_seed(0)  # every rank builds the same full batch, then keeps its block
    full = tn.randn([total, 6, 6, 6, 6], ranks_tt=5, batch=True, dtype=torch.float64)
    lo, hi = shard_range(total, world, rank)
    out = round_tt_sharded([c[lo:hi] for c in full.cores], rmax=3, algorithm="svd")
    # the pipelined form used by bench.py: async gather with sizes known up front, no merge copy
    from tntorch_amd.dist_batch import gather_batch, shard_range as sr

    t = tn.Tensor([c[lo:hi].clone() for c in full.cores], batch=True)
    t.round_tt(rmax=3, algorithm="svd")
    sizes = [sr(total, world, r)[1] - sr(total, world, r)[0] for r in range(world)]
    h = gather_batch(t, dst=0, sizes=sizes, async_op=True)
    parts = h.wait()
    if rank == 0:
        assert len(parts) == world and [p.cores[0].shape[0] for p in parts] == sizes
        merged = [torch.cat([p.cores[k] for p in parts]) for k in range(len(out.cores))]
        assert all(torch.equal(a, b) for a, b in zip(merged, out.cores))
        q.put([c.numpy() for c in out.cores])
    else:
        assert out is None and parts is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [6, 5])  # even and ragged split
def test_gloo_world2_matches_single_process(total):
    import tntorch_amd as tn
    from tntorch_amd.dist_batch import shard_range

    assert shard_range(5, 2, 0) == (0, 3) and shard_range(5, 2, 1) == (3, 5)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    torch.manual_seed(0)
    full = tn.randn([total, 6, 6, 6, 6], ranks_tt=5, batch=True, dtype=torch.float64)
    full.round_tt(rmax=3)
    for a, b in zip(got, full.cores):
        assert a.shape == tuple(b.shape)
        assert abs(torch.from_numpy(a) - b).max() < 1e-12


def test_pack_unpack_roundtrip():
    from tntorch_amd.dist_batch import pack_cores, unpack_cores

    cores = [torch.rand(3, 1, 4, 2), torch.rand(3, 2, 4, 5), torch.rand(3, 5, 4, 1)]
    flat = pack_cores(cores)
    back = unpack_cores(flat, [c.shape for c in cores])
    assert all(torch.equal(a, b) for a, b in zip(cores, back))


def test_pack_cores_zero_copy_when_back_to_back():
    """Cores that already lie back to back in one storage (how the device round_tt lays out a large batch) are
    packed as a view; anything else is concatenated."""
    from tntorch_amd.dist_batch import pack_cores

    flat = torch.arange(100.0)
    cores = [flat[0:24].view(2, 3, 2, 2), flat[24:60].view(2, 2, 3, 3), flat[60:100].view(2, 5, 2, 2)]
    p = pack_cores(cores)
    assert p.data_ptr() == flat.data_ptr() and torch.equal(p, flat)
    off = torch.arange(120.0)
    cores = [off[10:34].view(2, 3, 2, 2), off[34:70].view(2, 2, 3, 3)]
    p = pack_cores(cores)
    assert p.data_ptr() == off[10:].data_ptr() and torch.equal(p, off[10:70])
    q = pack_cores([c.clone() for c in cores])                  # separate storages: copy
    assert q.data_ptr() != off[10:].data_ptr() and torch.equal(q, off[10:70])
    gap = [off[10:34].view(2, 3, 2, 2), off[40:76].view(2, 2, 3, 3)]  # same storage, not adjacent: copy
    assert torch.equal(pack_cores(gap), torch.cat([off[10:34], off[40:76]]))


def _sched_worker(rank, world, port, q):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tntorch_amd as tn
    from tntorch_amd.dist_batch import GatherSchedule, shard_range

    torch.manual_seed(0)
    full = tn.randn([5, 6, 6, 6], ranks_tt=4, batch=True, dtype=torch.float64)
    lo, hi = shard_range(5, world, rank)
    sizes = [shard_range(5, world, r)[1] - shard_range(5, world, r)[0] for r in range(world)]
    report = {}
    for mode in GatherSchedule.MODES:
        sched = GatherSchedule(mode, sizes=sizes, dst=0)
        for k in range(3):  # three "steps": the same shard scaled by k + 1, rounded
            t = tn.Tensor([c[lo:hi] * (k + 1.0) if i == 0 else c[lo:hi].clone() for i, c in enumerate(full.cores)], batch=True)
            t.round_tt(rmax=2)
            sched.after_step(t)
        parts = sched.drain()
        assert sched.drain() is None                     # nothing is owed twice
        if rank == 0 and mode != "none":
            merged = torch.cat([p.cores[0] for p in parts])
            report[mode] = (sched.gathers, [p.cores[0].shape[0] for p in parts], merged.numpy())
        else:
            assert parts is None
            report[mode] = (sched.gathers, None, None)
        dist.barrier()
    if rank == 0:
        q.put(report)
    dist.destroy_process_group()


def test_gather_schedule_policies_gloo_world2():
    """bench.py's `--gather end | step | none` control flow (GatherSchedule) on two gloo ranks: `end` issues ONE collective and
    delivers the LAST step, `step` one per step (at most one in flight) and also ends with the last step, `none` none."""
    import tntorch_amd as tn

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sched_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    rep = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert rep["end"][0] == 1 and rep["step"][0] == 3 and rep["none"][0] == 0
    assert rep["end"][1] == rep["step"][1] == [3, 2]
    torch.manual_seed(0)
    full = tn.randn([5, 6, 6, 6], ranks_tt=4, batch=True, dtype=torch.float64)
    last = tn.Tensor([c * 3.0 if i == 0 else c.clone() for i, c in enumerate(full.cores)], batch=True)
    last.round_tt(rmax=2)
    for mode in ("end", "step"):
        assert abs(torch.from_numpy(rep[mode][2]) - last.cores[0]).max() < 1e-12
    with pytest.raises(ValueError):
        from tntorch_amd.dist_batch import GatherSchedule
        GatherSchedule("sometimes")
