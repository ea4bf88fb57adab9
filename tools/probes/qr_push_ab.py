"""Level-0 fused push + factor launch of the metric's shape (packed items: R of numerical rank 32 of 64, core = blockdiag(g, g)) under a
knob: `python tools/probes/qr_push_ab.py KNOB v0,v1,... [B]` prints the launch time (both levels, HIP events, best of 5) and the cycle
stamps of a few steady-state blocks per knob value (round 6: TTR_KNOB_QR_STAGGER = 16, TTR_KNOB_QR_PUSH = 17)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip as h

L = h.lib()
knob = int(sys.argv[1]) if len(sys.argv) > 1 else 16
vals = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
torch.manual_seed(0)
top = torch.triu(torch.randn(B, 32, 64, device="cuda"))
Rm = torch.cat([top, 1e-8 * torch.triu(torch.randn(B, 32, 64, device="cuda"), diagonal=32)], dim=1)
g = torch.randn(B, 32, 64, 32, device="cuda")
z = torch.zeros_like(g)
core = torch.cat([torch.cat([g, z], dim=-1), torch.cat([z, g], dim=-1)], dim=1).contiguous()
del g, z
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
ref = None
for rnd in range(2):   # two rounds: A/B/A/B
    for v in vals:
        h.set_knob(knob, v)
        f = h.qr_factor_pushed(Rm, core); torch.cuda.synchronize()
        if ref is None:
            ref = f.R.clone()
        same = bool(torch.equal(ref, f.R))
        if rnd == 0:
            import hashlib
            print(f"    sha256(R) {hashlib.sha256(f.R.cpu().numpy().tobytes()).hexdigest()[:16]}  lib {os.environ.get('TTR_LIB_PATH', 'default')}")
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); h.qr_factor_pushed(Rm, core); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"knob {knob} = {v}: launch (both levels) best {min(ts):.3f} ms  median {sorted(ts)[2]:.3f} ms  R bit-identical to the first variant: {same}", flush=True)
        if rnd == 0:
            for bx, by in ((1, B // 4), (2, B // 2), (3, B // 2 + 1)):
                h.set_knob(h.KNOB_QR_STAMP_BX, bx); h.set_knob(h.KNOB_QR_STAMP_BY, by)
                L.ttr_debug_set_qr_stamps(buf.data_ptr()); buf.zero_()
                h.qr_factor_pushed(Rm, core); torch.cuda.synchronize()
                L.ttr_debug_set_qr_stamps(None)
                st = [x for x in buf.cpu().tolist() if x != 0]
                d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
                print(f"    block ({bx}, {by}): total {st[-1] - st[0] if st else 0}  push {d[0] if d else 0}  rest {d[1:]}")
            h.set_knob(h.KNOB_QR_STAMP_BX, 0); h.set_knob(h.KNOB_QR_STAMP_BY, 0)
h.set_knob(knob, vals[0])
