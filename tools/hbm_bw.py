"""What stock streaming kernels reach on this box (4 GiB tensors): the practical HBM ceilings beside the 8 TB/s datasheet figure."""
import torch, time
x = torch.empty(1 << 30, dtype=torch.float32, device='cuda').normal_()   # 4 GiB
y = torch.empty_like(x)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
s = t(lambda: x.sum())
c = t(lambda: y.copy_(x))
f = t(lambda: y.fill_(1.0))
print("read (sum) %.2f TB/s   copy (r+w) %.2f TB/s   fill (write) %.2f TB/s" % (x.numel()*4/s/1e12, 2*x.numel()*4/c/1e12, x.numel()*4/f/1e12))
