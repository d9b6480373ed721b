import sys, os, math
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch
import test_gpu_convergence as T
from rise_sdf_amd.synthetic import make_dataset
dev = torch.device('cuda:0')
ds = make_dataset(n_views=6, W=64, H=64, seed=3, device=dev)
batches = T._batches(dev, ds, T.STEPS, T.N_RAYS, seed=11)
held = T._batches(dev, ds, 1, 2048, seed=99)[0]
heldout = (held[0], held[1])
for eps in (1e-12, 1e-8):
    import torch.optim
    orig = torch.optim.Adam
    class A(orig):
        def __init__(self, params, **kw):
            kw['eps'] = eps
            super().__init__(params, **kw)
    torch.optim.Adam = A
    la, pa, _ = T._train_hip(dev, 'fp32', batches, heldout)
    lb, pb, _ = T._train_hip(dev, 'fp32', batches, heldout)
    lc, pc, _ = T._train_hip(dev, 'bf16', batches, heldout)
    lo, po = T._train_oracle(T._model(dev, 'fp32'), batches, heldout)
    torch.optim.Adam = orig
    g = lambda x, y: [abs(a - b) / abs(b) for a, b in zip(x, y)]
    print('eps', eps, 'PSNR hipA %.3f hipB %.3f bf16 %.3f oracle %.3f' % (pa, pb, pc, po))
    for name, gg in (('A-B', g(la, lb)), ('A-oracle', g(la, lo)), ('bf16-A', g(lc, la))):
        print('  gap', name, ['%.1e' % v for v in gg[:12]], 'mean %.3e max %.3e' % (sum(gg) / len(gg), max(gg)))
    w = lambda l: sum(l[-20:]) / 20
    print('  mean loss last 20: A %.4f B %.4f bf16 %.4f oracle %.4f' % (w(la), w(lb), w(lc), w(lo)))
