import sys, math, torch
sys.path.insert(0, '.')
import oracle, tntorch_amd as tn
from tntorch_amd import _hip, _hipops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.manual_seed(seed)
g = oracle.tt_randn([64]*8, 32, dtype=torch.float32, batch_size=B)
inp = oracle.tt_add(g, g, batch=True)
c = [x.cuda() for x in inp]
# L2R
for mu in range(7):
    _hipops.left_orthogonalize(c, mu)
for mu in range(7):
    L = c[mu].reshape(B, -1, c[mu].shape[-1]).double()
    e = (L.transpose(1,2)@L - torch.eye(L.shape[-1], device='cuda', dtype=torch.float64)).abs().amax(dim=(1,2))
    print('L2R core', mu, 'orth err per item', [f'{x:.1e}' for x in e.tolist()])
# truncation of last core
for alg in ('svd', 'eig'):
    M = c[7].reshape(B, 64, 64)
    t = _hipops.truncate(M, None, 32, False, alg, True)
    R = t.right.double()
    e = (R@R.transpose(1,2) - torch.eye(32, device='cuda', dtype=torch.float64)).abs().amax(dim=(1,2))
    rec = (t.left_scaled().double() @ R - M.double()).norm(dim=(1,2)) / M.double().norm(dim=(1,2))
    print(alg, 'right orth err', [f'{x:.1e}' for x in e.tolist()], 'recon', [f'{x:.1e}' for x in rec.tolist()])
    print('  sigma[0..3], sigma[30..33]', t.colscale[0, :4].tolist(), t.colscale[0, 30:34].tolist())
out = _hipops.round_tt([x.cuda() for x in inp], 1e-14, [32]*7, 'svd', True)
for k in range(1, 8):
    Rm = out[k].reshape(B, out[k].shape[1], -1).double()
    e = (Rm@Rm.transpose(1,2) - torch.eye(Rm.shape[1], device='cuda', dtype=torch.float64)).abs().amax(dim=(1,2))
    print('round core', k, [f'{x:.1e}' for x in e.tolist()])
