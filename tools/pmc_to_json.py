#!/usr/bin/env python3
"""Turn the rocprofv3 passes of tools/profile_round.sh into profiles/pmc_latest.json (what bench.py reports as
`roofline.traffic` / `mfma_util`) and a readable per-kernel table.
    python tools/pmc_to_json.py <dir with pmc_*/ and kt/> <steps in the profiled run> <B> > summary.txt
HBM bytes: FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request of a wide
coalesced stream (MI355X_MICROARCH.md, HBM section): the x2-corrected value is used and both are printed.
MFMA utilisation: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs) -- rocprofv3's own MfmaUtil expression --
normalised by the same ratio measured on a kernel that issues nothing but back-to-back v_mfma_f32_16x16x4_f32
(tools/microbench.hip `m`), so that 1.0 means "the matrix pipe never idles"."""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kind_of(name):
    if "qr_factor_kernel" in name or "r_expo_kernel" in name:   # (r_expo: the normalisation of a single-level pushed factorisation, timed as qr_factor)
        return "qr_factor"
    if "qr_apply_kernel" in name:
        return "qr_apply"
    if "colgram_kernel" in name:
        return "colgram"
    if "colproject_kernel" in name:
        return "colproject"
    if "krp_contract" in name:
        return "krp_contract"
    if "rotgram_kernel" in name:
        return "rowgram" if re.search(r"rotgram_kernel<\w+, true>", name) else "rotgram"
    if "project_kernel" in name and "colproject" not in name:
        return "project"
    if "eigh_" in name:
        return "eigh"
    if "gemm_kernel" in name or "gemm_big_kernel" in name or "splitk_reduce" in name or "bj_apply_kernel" in name:
        return "gemm"
    if "ttr::" in name:
        return "misc"
    return None


sys.path.insert(0, ROOT)
from bench import kind_shas, source_sha  # noqa: E402  (one definition of the hashes for writer and reader)


def read_pass(d):
    """-> {kernel name: {counter: [values per dispatch]}}, {kernel name: [durations ns]}"""
    vals = defaultdict(lambda: defaultdict(list))
    durs = defaultdict(list)
    for cc in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(cc)):
            vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                durs[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return vals, durs


def main():
    base, steps, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    per_kind = defaultdict(lambda: defaultdict(float))
    per_kernel = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    for d in sorted(glob.glob(os.path.join(base, "pmc_*"))):
        if d.endswith("_calib"):
            continue
        vals, durs = read_pass(d)
        for name, cs in vals.items():
            k = kind_of(name)
            if k is None:
                continue
            short = re.sub(r"\(.*$", "", name).replace("void ", "")[:70]
            for c, v in cs.items():
                per_kind[k][c] += sum(v)
                per_kernel[short][c] += sum(v)
                calls[short] = max(calls[short], len(v))
    calib = None
    cd = os.path.join(base, "pmc_mfma_calib")
    if os.path.isdir(cd):
        vals, _ = read_pass(cd)
        for name, cs in vals.items():
            if ("mfma_kernel" in name or "mfma64_kernel" in name) and cs.get("GRBM_GUI_ACTIVE"):
                calib = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(cs["GRBM_GUI_ACTIVE"]) * 1024.0)
    out = {"_source": f"tools/profile_round.sh ({steps} single-stream steps at B = {B}; separate --pmc passes)", "_batch": B,
           "source_sha": source_sha(), "kind_sha": kind_shas(), "mfma_busy_ratio_of_pure_mfma_kernel": calib}
    print(f"# PMC summary ({steps} steps, B = {B}); MFMA-busy ratio of a pure-MFMA kernel: {calib}")
    print(f"{'kind':10s} {'fetch_raw_GB/step':>18s} {'fetch_x2_GB/step':>18s} {'write_GB/step':>14s} {'hbm_GB/step':>12s} {'mfma_busy/(gui*1024)':>22s} {'mfma_util':>10s} {'mfma_TFLOP/step':>16s}")
    for k, c in sorted(per_kind.items()):
        fetch = c.get("FETCH_SIZE", 0.0) * 1024 / steps
        write = c.get("WRITE_SIZE", 0.0) * 1024 / steps
        entry = {}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            entry["hbm_bytes_per_step"] = 2 * fetch + write
            entry["fetch_bytes_raw_per_step"] = fetch
            entry["write_bytes_per_step"] = write
        ratio = util = None
        if c.get("GRBM_GUI_ACTIVE"):
            ratio = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] * 1024.0)
            util = ratio / calib if calib else ratio
            entry["mfma_busy_ratio"] = ratio
            entry["mfma_util"] = util
        mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", c.get("SQ_INSTS_VALU_MFMA_MOPS_F64"))
        if mops is not None:
            entry["mfma_flops_per_step"] = mops * 512 / steps
        out[k] = entry
        print(f"{k:10s} {fetch/1e9:18.3f} {2*fetch/1e9:18.3f} {write/1e9:14.3f} {(2*fetch+write)/1e9:12.3f} "
              f"{'' if ratio is None else format(ratio, '22.4f')} {'' if util is None else format(util, '10.3f')} "
              f"{'' if mops is None else format(mops*512/steps/1e12, '16.3f')}")
    print()
    print(f"{'kernel':72s} {'calls':>6s} {'fetch_x2_MB/call':>17s} {'write_MB/call':>14s} {'mfma_ratio':>11s}")
    for name, c in sorted(per_kernel.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
        n = max(1, calls[name])
        ratio = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] * 1024.0) if c.get("GRBM_GUI_ACTIVE") else float("nan")
        print(f"{name:72s} {n:6d} {2*c.get('FETCH_SIZE',0)*1024/n/1e6:17.2f} {c.get('WRITE_SIZE',0)*1024/n/1e6:14.2f} {ratio:11.4f}")
    json.dump(out, open(os.path.join(base, "pmc_latest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
