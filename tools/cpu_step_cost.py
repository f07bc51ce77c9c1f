"""Host-side cost of one Trainer.train_step: with 32 rays x 16 samples the device work is negligible, so the time per step is
the Python + launch overhead that must stay below the device time of a real step (5 ms) for the sync-free loop to run ahead."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
import torch
import bench

if __name__ == "__main__":
    bench.R_PER_GPU, bench.N_SAMPLES = 32, 16
    dev = torch.device("cuda", 0)
    trainer, net = bench.build_trainer(dev, 1, "--aux" in sys.argv)
    data = bench.synthetic_batch(dev)
    for i in range(10):
        trainer.train_step(data, it=i, epoch=0, scheduling_start=10000, render_path=None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for i in range(n):
        trainer.train_step(data, it=10 + i, epoch=0, scheduling_start=10000, render_path=None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host time per step %.3f ms (enqueue only), %.3f ms including the final drain; cpu count %d" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, os.cpu_count()))
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(50):
        trainer.train_step(data, it=200 + i, epoch=0, scheduling_start=10000, render_path=None)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)
