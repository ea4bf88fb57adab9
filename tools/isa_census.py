"""ISA census of the hot kernels (no GPU needed): compiles tntorch_amd/csrc/*.hip to gfx950 assembly and counts, per kernel instance,
the global memory instructions by width (global_load_dword / x2 / x3 / x4, global_load_lds_*, global_store_dword / x2 / x4), LDS reads /
writes, MFMAs, scratch (spill) instructions and the code size in instructions.
    python tools/isa_census.py > profiles/r06_isa_census.txt"""
import collections
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tntorch_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HOT = ("qr_factor_kernel<float, 4, true, 8, true>", "qr_factor_kernel<float, 4, false, 8, true>", "qr_apply_kernel<float, 4, 2, 8>",
       "project_kernel<float>", "rotgram_kernel<float, true>", "rotgram_kernel<float, false>", "eigh_tridiag_kernel<float, true, 32, 3>",
       "eigh_tridiag_kernel<float, true, 64, 0>", "eigh_jacobi_kernel<float, true, 256>", "colgram_kernel<float, false>",
       "colproject_kernel<float, false>", "gemm_big_kernel<float>", "pack_flags_kernel<float>", "qr_factor_kernel<double, 4, true, 8, true>",
       "qr_apply_kernel<double, 4, 2, 8>")
tmp = tempfile.mkdtemp()
procs = []
for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
    base = os.path.basename(src)[:-4]
    extra = ["-fno-slp-vectorize"] if base == "ttr_eigh" else []
    out = os.path.join(tmp, base + ".s")
    procs.append((base, out, subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only"] + extra +
                                               [src, "-o", out], stderr=subprocess.DEVNULL, cwd=CSRC)))
COLS = ["global_load_dword", "global_load_dwordx2", "global_load_dwordx3", "global_load_dwordx4", "global_load_lds", "global_store_dword",
        "global_store_dwordx2", "global_store_dwordx4", "ds_read", "ds_write", "v_mfma", "scratch", "instrs"]
rows = []
for base, out, p in procs:
    p.wait()
    cur, cnt = None, None
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, cnt = m.group(1), collections.Counter()
            rows.append((base, cur, cnt))
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        t = line.split()
        if not t or t[0].startswith((";", ".", "//")) or t[0].endswith(":"):
            continue
        op = t[0]
        cnt["instrs"] += 1
        if op.startswith("global_load_lds"):
            cnt["global_load_lds"] += 1
        elif op in ("global_load_dword", "global_load_dwordx2", "global_load_dwordx3", "global_load_dwordx4", "global_store_dword",
                    "global_store_dwordx2", "global_store_dwordx4"):
            cnt[op] += 1
        elif op.startswith("ds_read"):
            cnt["ds_read"] += 1
        elif op.startswith("ds_write"):
            cnt["ds_write"] += 1
        elif op.startswith("v_mfma"):
            cnt["v_mfma"] += 1
        elif op.startswith("scratch_"):
            cnt["scratch"] += 1
names = subprocess.run(["c++filt"] + [r[1] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("# ISA census of the hot kernels (hipcc -S --offload-arch=gfx950 of tntorch_amd/csrc; tools/isa_census.py): static instruction counts per")
print("# kernel instance.  gl = global_load_dword{,x2,x3,x4}, lds-dma = global_load_lds_*, gs = global_store_dword{,x2,x4}.")
print(f"{'file':10s} {'gl':>5s} {'glx2':>5s} {'glx3':>5s} {'glx4':>5s} {'ldsdma':>6s} {'gs':>5s} {'gsx2':>5s} {'gsx4':>5s} {'ds_rd':>6s} {'ds_wr':>6s} {'mfma':>6s} {'scratch':>7s} {'instrs':>7s}  kernel")
for (base, mangled, cnt), n in zip(rows, names):
    n = re.sub(r"\(.*$", "", n).replace("void ", "").replace("ttr::", "")
    if "--all" not in sys.argv and not any(n.endswith(h) or n == h for h in HOT):
        continue
    print(f"{base[4:]:10s} " + " ".join(f"{cnt[c]:{w}d}" for c, w in zip(COLS, (5, 5, 5, 5, 6, 5, 5, 5, 6, 6, 6, 7, 7))) + f"  {n}")
