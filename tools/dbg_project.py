import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tntorch_amd import _hip as h
torch.manual_seed(0)
for (B, R, n, ro) in ((3, 64, 2048, 32), (1, 64, 64, 32), (2, 64, 2048, 64), (1, 16, 64, 16)):
    M = torch.randn(B, R, n).cuda()
    V1 = torch.linalg.qr(torch.randn(B, R, R))[0].cuda().contiguous()
    V2 = torch.linalg.qr(torch.randn(B, R, R))[0].cuda().contiguous()
    sig = (torch.rand(B, R) + 0.5).cuda()
    for scale in (True, False):
        for useV1 in (True, False):
            right, left = h.project(M, V1 if useV1 else None, V2, sig, ro, scale_right=scale)
            U = (V1.double() @ V2.double()[:, :, :ro]) if useV1 else V2.double()[:, :, :ro]
            want_l = U * sig.double()[:, None, :ro] if scale else U
            want_r = (U.transpose(1, 2) @ M.double())
            if scale:
                want_r = want_r / sig.double()[:, :ro, None]
            el = (left.double() - want_l).abs()
            er = (right.double() - want_r).abs()
            bad = (el > 1e-4).nonzero()
            print(f"B={B} R={R} n={n} ro={ro} scale={scale} V1={useV1}: left err {el.max().item():.2e} ({len(bad)} bad) right err {er.max().item():.2e}")
            if len(bad):
                ks = sorted(set(bad[:, 1].tolist())); iis = sorted(set(bad[:, 2].tolist())); bs = sorted(set(bad[:, 0].tolist()))
                print("   bad batch", bs, "rows", ks[:20], "...", "cols", iis[:40])
                print("   sample ours", left[bad[0][0], bad[0][1], bad[0][2]].item(), "want", want_l[bad[0][0], bad[0][1], bad[0][2]].item())
