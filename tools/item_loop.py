import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
import torch, bench
dev = torch.device("cuda", 0)
for aux in (False, True):
    trainer, net = bench.build_trainer(dev, 1, aux)
    data = bench.synthetic_batch(dev)
    for mode in ("no item", "3 x item"):
        for i in range(20):
            ld = trainer.train_step(data, it=i + 1, epoch=0, scheduling_start=10000, render_path=None)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 200
        for i in range(n):
            ld = trainer.train_step(data, it=i + 21, epoch=0, scheduling_start=10000, render_path=None)
            if mode != "no item":
                a = ld['l2_mean'].item(); b = ld['loss_pc'].item(); c = ld['loss_rgb_s'].item()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print("aux=%s %-9s %.3f ms/step  types: %s" % (aux, mode, (t1 - t0) / n * 1e3, {k: type(ld[k]).__name__ for k in ('l2_mean', 'loss_pc', 'loss_rgb_s', 'loss')}))
