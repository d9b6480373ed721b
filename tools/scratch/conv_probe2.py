import sys, os, math
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch
import test_gpu_convergence as T
from rise_sdf_amd.synthetic import make_dataset
from rise_sdf_amd.loss import loss_tail
dev = torch.device('cuda:0')
ds = make_dataset(n_views=6, W=64, H=64, seed=3, device=dev)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 800
batches = T._batches(dev, ds, NS, 256, seed=11)
held = T._batches(dev, ds, 1, 4096, seed=99)[0]
for prec, sdfp, lrs in (('bf16', 'fp32', 0.4), ('bf16', 'fp32', 0.4), ('fp32', 'bf16', 0.4), ('fp32', 'bf16', 0.4), ('fp32', 'fp32', 0.4)):
    model = T._model(dev, prec, sdfp)
    opt = torch.optim.Adam([{"params": getattr(model, k).parameters(), "lr": lr * lrs} for k, lr in T.LRS.items()], betas=(0.9, 0.999), eps=1e-12)
    ps = []
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda k: max(0.0, 1.0 - k / NS))
    for i, (rays, rgb, fg, u) in enumerate(batches):
        out = model.forward_(rays, stratified_u=u)
        loss, _ = loss_tail(out, {"rgb": rgb, "fg_mask": fg}, T.LAMBDAS)
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step(); sched.step()
        if (i + 1) % 100 == 0:
            model.eval()
            with torch.no_grad():
                ps.append(T._psnr(model(held[0])["comp_rgb_full"], held[1]))
            model.train()
    print('tex', prec, 'sdf', sdfp, lrs, ' '.join('%.2f' % p for p in ps))
