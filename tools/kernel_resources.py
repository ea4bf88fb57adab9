"""Compile-time resources of every kernel instance (VGPRs, occupancy, spills, scratch, static LDS): compiles each translation unit
of tntorch_amd/csrc with -Rpass-analysis=kernel-resource-usage (no GPU needed) and prints one table.
    python tools/kernel_resources.py > profiles/r05_kernel_resources.txt"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tntorch_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
tmp = tempfile.mkdtemp()
procs = []
for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
    base = os.path.basename(src)[:-4]
    log = open(os.path.join(tmp, base + ".txt"), "w")
    procs.append((base, log, subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o",
                                                os.path.join(tmp, base + ".o"), "-Rpass-analysis=kernel-resource-usage"],
                                               stderr=log, cwd=CSRC)))
rows = []
for base, log, p in procs:
    p.wait()
    log.close()
    cur = None
    for line in open(log.name):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1), "file": base}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("spill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and key not in cur:
                cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("# Compile-time resources of every kernel instance of this build (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; tools/kernel_resources.py):")
print("# VGPRs (512 per SIMD lane: occ = waves per SIMD the register count allows), spilled VGPRs / scratch bytes per lane, static LDS bytes per")
print("# workgroup (dynamic LDS is set at launch and not shown).")
print(f"{'file':12s} {'VGPR':>5s} {'occ':>4s} {'spill':>6s} {'scratch':>8s} {'LDS':>7s}  kernel")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*$", "", n).replace("void ", "")
    print(f"{r['file']:12s} {r.get('vgpr', 0):5d} {r.get('occ', 0):4d} {r.get('spill', 0):6d} {r.get('scratch', 0):8d} {r.get('lds', 0):7d}  {n}")
