"""Practical memory ceilings on this box (torch ops as probes): copy, row gather, row scatter."""
import torch, time
dev = "cuda:0"
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for mb in (92, 184, 368, 1024):
    x = torch.empty(mb * 1024 * 1024 // 4, device=dev, dtype=torch.float32).normal_()
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    print(f"copy {mb} MB: {2*mb*1.048576e6/t/1e12:.2f} TB/s (r+w)  {t*1e6:.1f} us")
    t = timeit(lambda: x.sum())
    print(f"read-only sum {mb} MB: {mb*1.048576e6/t/1e12:.2f} TB/s  {t*1e6:.1f} us")
    t = timeit(lambda: y.fill_(1.0))
    print(f"write-only fill {mb} MB: {mb*1.048576e6/t/1e12:.2f} TB/s  {t*1e6:.1f} us")
n = 8 * 60032
for rowf in (16, 32, 64):
    src = torch.randn(n, rowf, device=dev)
    perm = torch.randperm(n, device=dev)
    out = torch.empty_like(src)
    t = timeit(lambda: torch.index_select(src, 0, perm, out=out))
    print(f"gather rows of {rowf*4} B x {n}: {2*src.numel()*4/t/1e12:.2f} TB/s (r+w) {t*1e6:.1f} us")
    t = timeit(lambda: out.index_copy_(0, perm, src))
    print(f"scatter rows of {rowf*4} B x {n}: {2*src.numel()*4/t/1e12:.2f} TB/s (r+w) {t*1e6:.1f} us")
