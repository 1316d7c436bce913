"""Achievable HBM bandwidth of plain streaming kernels on this box (torch ops, HIP events): the yardstick for the BatchNorm streams."""
import torch
dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e-3


for mb in (206, 411, 822, 1644):
    n = mb * (1 << 20) // 2
    x = torch.randn(n, device=dev, dtype=torch.float16)
    y = torch.empty_like(x)
    z = torch.randn(n, device=dev, dtype=torch.float16)
    t = timeit(lambda: x.sum(dtype=torch.float32))
    print(f"{mb:5d} MB  sum (read)            {2 * n / t / 1e12:6.2f} TB/s")
    t = timeit(lambda: torch.dot(x, z))
    print(f"{mb:5d} MB  dot (2 reads)         {4 * n / t / 1e12:6.2f} TB/s")
    t = timeit(lambda: y.copy_(x))
    print(f"{mb:5d} MB  copy (read + write)   {4 * n / t / 1e12:6.2f} TB/s")
    t = timeit(lambda: torch.add(x, z, out=y))
    print(f"{mb:5d} MB  add (2 reads + write) {6 * n / t / 1e12:6.2f} TB/s")
    t = timeit(lambda: y.fill_(1.0))
    print(f"{mb:5d} MB  fill (write)          {2 * n / t / 1e12:6.2f} TB/s")
