"""``tntorch_amd.patch(tntorch)`` on the reference's own class: the bodies of the reference's
tests/test_round.py:7-68 run through the patched methods.  CPU box: the real reference from /root/reference, host
mirror underneath.  GPU box (no /root/reference there): a minimal stand-in class with the reference's attributes,
device cores, HIP kernels underneath."""
import os
import sys
import types

import numpy as np
import pytest
import torch

import oracle
import tntorch_amd as tna

REF = "/root/reference"


@pytest.fixture()
def ref_tn():
    if not os.path.isdir(os.path.join(REF, "tntorch")):
        pytest.skip("reference tree not present")
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import tntorch as tn
    saved = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)  # tests/test_round.py:4
    undo = tna.patch(tn)
    yield tn
    undo()
    torch.set_default_dtype(saved)


def test_patch_installs_and_uninstalls(ref_tn):
    tn = ref_tn
    assert hasattr(tn.Tensor.round_tt, "_tntorch_amd_original") and tn.truncated_svd is tna.truncated_svd
    first = tn.Tensor.round_tt
    undo2 = tna.patch(tn)  # idempotent: an already patched module is left alone and the SAME undo comes back
    assert tn.Tensor.round_tt is first
    undo2()
    assert not hasattr(tn.Tensor.round_tt, "_tntorch_amd_original") and tn.truncated_svd is not tna.truncated_svd
    undo2()  # a second undo is a no-op
    assert not hasattr(tn.Tensor.round_tt, "_tntorch_amd_original")
    undo3 = tna.patch(tn)  # (the fixture's undo is the first one: a no-op by now, so restore here)
    assert hasattr(tn.Tensor.round_tt, "_tntorch_amd_original")
    undo3()
    tna.patch(tn)  # leave the module patched for the fixture's teardown ... which holds the FIRST undo (no-op): undo below
    tna._patch._ACTIVE[id(tn)][1]()


def test_reference_orthogonalization_body(ref_tn):
    """tests/test_round.py:7-19 through the patched methods (20 random trains instead of 100)."""
    tn = ref_tn
    np.random.seed(0)
    torch.manual_seed(0)
    for _ in range(20):
        gt = tn.rand(np.random.randint(1, 8, np.random.randint(2, 6)))
        t = gt.clone()
        held = list(t.cores)
        snap = [c.clone() for c in held]
        assert tn.relative_error(gt, t) <= 1e-7
        t.left_orthogonalize(0)
        assert tn.relative_error(gt, t) <= 1e-7
        t.right_orthogonalize(t.dim() - 1)
        assert tn.relative_error(gt, t) <= 1e-7
        t.orthogonalize(np.random.randint(t.dim()))
        assert tn.relative_error(gt, t) <= 1e-7
        assert all(torch.equal(a, b) for a, b in zip(held, snap))  # rebinding: the old core tensors were never written


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_reference_truncated_svd_body(ref_tn, alg):
    """tests/test_round.py:22-39: batch == loop."""
    tn = ref_tn
    torch.manual_seed(1)
    gt = torch.rand((2, 32, 32))
    u, v = tn.truncated_svd(gt, batch=True, algorithm=alg)
    for i in range(len(gt)):
        u1, v1 = tn.truncated_svd(gt[i], batch=False, algorithm=alg)
        assert torch.allclose(u1, u[i]) and torch.allclose(v1, v[i])


@pytest.mark.parametrize("alg,tol", [("svd", 1e-4), ("eig", 1e-7)])
def test_reference_round_tt_body(ref_tn, alg, tol):
    """tests/test_round.py:41-59 (10 trains instead of 100)."""
    tn = ref_tn
    np.random.seed(2)
    torch.manual_seed(2)
    for _ in range(10):
        gt = tn.rand(np.random.randint(1, 8, np.random.randint(8, 10)), ranks_tt=np.random.randint(1, 10))
        gt.round_tt(1e-8, algorithm=alg)
        t = gt + gt
        t.round_tt(1e-8, algorithm=alg)
        assert tn.relative_error(gt, t / 2) <= tol
        if alg == "svd":
            assert max(gt.ranks_tt) == max(t.ranks_tt)
        t2 = tn.round_tt(gt + gt, eps=1e-8, algorithm=alg)  # free function: clone + patched method (round.py:7-19)
        assert tn.relative_error(gt, t2 / 2) <= tol


def test_reference_round_tucker_body(ref_tn):
    """tests/test_round.py:62-68 (5 tensors instead of 100)."""
    tn = ref_tn
    np.random.seed(3)
    torch.manual_seed(3)
    for _ in range(5):
        eps = np.random.rand() ** 2
        gt = tn.rand([32] * 4, ranks_tt=8, ranks_tucker=8)
        t = gt.clone()
        t.round_tucker(eps=eps)
        assert tn.relative_error(gt, t) <= eps


def test_patched_result_equals_unpatched(ref_tn):
    """Same input through the reference's own code and through the patched class: identical ranks, same tensor."""
    tn = ref_tn
    torch.manual_seed(4)
    g = tn.rand([6, 7, 5, 8, 6], ranks_tt=4)
    t = g + g
    a = t.clone()
    a.round_tt(eps=1e-8)                                        # patched (host mirror)
    b = t.clone()
    tn.Tensor.round_tt._tntorch_amd_original(b, eps=1e-8)       # the reference's own method
    assert list(a.ranks_tt) == list(b.ranks_tt)
    assert (a.torch() - b.torch()).norm() / b.torch().norm() < 1e-12


# ------------------------------------------------------------------ GPU box: stand-in for the reference class
class _RefLike:
    """The attributes / construction the reference's Tensor exposes on this path (tensor.py:107-209)."""

    def __init__(self, cores, Us=None, idxs=None, batch=False):
        self.cores = list(cores)
        self.Us = [None] * len(cores) if Us is None else list(Us)
        self.idxs = idxs
        self.batch = batch

    def clone(self):
        return _RefLike([c.clone() for c in self.cores], [None if U is None else U.clone() for U in self.Us], self.idxs, self.batch)

    # unpatched placeholders (the reference's own methods would sit here)
    def round_tt(self, *a, **k):
        raise AssertionError("unpatched")

    orthogonalize = left_orthogonalize = right_orthogonalize = round_tucker = round = factor_orthogonalize = round_tt


@pytest.mark.gpu
def test_patch_on_device_cores():
    mod = types.SimpleNamespace(Tensor=_RefLike, truncated_svd=None)
    undo = tna.patch(mod)
    try:
        torch.manual_seed(5)
        g = oracle.tt_randn([8, 9, 7, 8, 6], 5, dtype=torch.float64)
        inp = oracle.tt_add(g, g)
        t = _RefLike([c.cuda() for c in inp])
        held = list(t.cores)
        snap = [c.clone() for c in held]
        t.round_tt(eps=1e-8)
        assert [c.shape[-1] for c in t.cores] == [5, 5, 5, 5, 1] and all(c.is_cuda for c in t.cores)
        ref = oracle.round_tt(inp, eps=1e-8)
        a = oracle.tt_to_dense([c.cpu() for c in t.cores])
        b = oracle.tt_to_dense(ref)
        assert (a - b).norm() / b.norm() < 1e-10
        assert all(torch.equal(x, y) for x, y in zip(held, snap))
        t2 = _RefLike([c.cuda() for c in inp])
        t2.orthogonalize(2)
        a2 = oracle.tt_to_dense([c.cpu() for c in t2.cores])
        assert (a2 - oracle.tt_to_dense(inp)).norm() / a2.norm() < 1e-12
        M = torch.rand(40, 90, dtype=torch.float64).cuda()
        L, R = mod.truncated_svd(M, eps=1e-3)
        assert ((L @ R) - M).norm() / M.norm() <= 1e-3
    finally:
        undo()
    with pytest.raises(AssertionError):
        _RefLike([torch.zeros(1, 2, 1)]).round_tt()


def test_reduce_tree_matches_reference_order():
    """tools.reduce (own implementation) builds the same tree and hands the operands over in the same order as
    tools.py:460-512: checked with a non-commutative combiner on 1..9 tensors against the unpatched reference."""
    if not os.path.isdir(os.path.join(REF, "tntorch")):
        pytest.skip("reference tree not present")
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import tntorch as tn
    torch.manual_seed(6)
    for count in range(1, 10):
        cores = [oracle.tt_randn([4, 5, 3], 2, dtype=torch.float64) for _ in range(count)]

        def comb(a, b):
            return a + b * 2.0  # order-sensitive: every tree shape / operand order gives different weights

        ref = tn.reduce([tn.Tensor([c.clone() for c in cs]) for cs in cores], comb, eps=1e-12)
        ours = tna.reduce((tna.Tensor([c.clone() for c in cs]) for cs in cores), comb, eps=1e-12)  # a generator works too
        assert (ours.torch() - ref.torch()).norm() / ref.torch().norm() < 1e-10, count
    with pytest.raises(ValueError):
        tna.reduce([], comb)
