"""Parity comparators shared by the CPU and GPU tests (gauge-invariant where needed).

Singular-vector signs (and rotations inside degenerate clusters) differ between LAPACK
and the Jacobi kernels, so cores are compared (i) through gauge-invariant quantities --
ranks, bond singular values, dense reconstruction -- and (ii) directly after the per-bond
+-1 sign gauge of ``oracle.gauge_align`` when the spectrum is separated.
"""
import json
import os

import numpy as np
import torch

import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_meta():
    with open(os.path.join(GOLDEN, "golden_meta.json")) as f:
        return json.load(f)


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    groups = {}
    for key in z.files:
        if "/" in key:
            g, k = key.split("/")
            groups.setdefault(g, {})[int(k)] = torch.from_numpy(z[key])
        else:
            groups[key] = torch.from_numpy(z[key])
    out = {}
    for g, v in groups.items():
        out[g] = [v[i] for i in range(len(v))] if isinstance(v, dict) else v
    return out


def to_list(cores, batch_index=None):
    cs = [c.detach().cpu() for c in cores]
    if batch_index is not None:
        cs = [c[batch_index] for c in cs]
    return cs


def ranks(cores):
    return oracle.tt_ranks(cores)


def dense(cores):
    return oracle.tt_to_dense([c.double() for c in cores])


def rel_diff(a, b):
    return (torch.norm(a.double() - b.double()) / torch.norm(b.double()).clamp_min(1e-300)).item()


def assert_tt_close(ours, ref, tol_dense, tol_sv=None, tol_cores=None, what=""):
    """ours/ref: lists of CPU cores of ONE tensor train."""
    assert ranks(ours) == ranks(ref), f"{what}: ranks {ranks(ours)} != {ranks(ref)}"
    d = rel_diff(dense(ours), dense(ref))
    assert d <= tol_dense, f"{what}: dense reconstruction differs by {d:.3e} > {tol_dense:.1e}"
    if tol_sv is not None:
        so = oracle.bond_singular_values(ours)
        sr = oracle.bond_singular_values(ref)
        for k, (a, b) in enumerate(zip(so, sr)):
            e = ((a - b).abs().max() / b.max()).item()
            assert e <= tol_sv, f"{what}: bond {k} singular values differ by {e:.3e} > {tol_sv:.1e}"
    if tol_cores is not None:
        al = oracle.gauge_align(ref, ours)
        for k, (a, b) in enumerate(zip(al, ref)):
            e = (a.double() - b.double()).abs().max().item() / max(b.abs().max().item(), 1e-300)
            assert e <= tol_cores, f"{what}: sign-gauged core {k} differs by {e:.3e} > {tol_cores:.1e}"


def analytic_128():
    """docs/tutorials/decompositions.ipynb cell 1."""
    X, Y, Z = np.meshgrid(range(128), range(128), range(128))
    return torch.tensor(np.sqrt(np.sqrt(X) * (Y + Z) + Y * Z**2) * (X + np.sin(Y) * np.cos(Z)), dtype=torch.float64)
