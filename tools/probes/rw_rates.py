"""Read-only, write-only and copy rates of plain torch kernels on planes of the metric's size (514 MB) -- what the plane cost model of DESIGN.md section 7
(63 us per plane read, 122 us per plane written) is made of: python tools/probes/rw_rates.py"""
import torch

dev = torch.device('cuda')
n = 5 * 50176 * 32 * 16
bufs = [torch.randn(n, device=dev) for _ in range(6)]
out = [torch.empty(n, device=dev) for _ in range(6)]


def timed(fn, reps=12):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


mb = n * 4 / 1e6
us = timed(lambda i: out[i % 6].fill_(1.0))
print(f'write only (fill)        {us:7.1f} us  {mb / us:6.2f} TB/s')
us = timed(lambda i: out[i % 6].zero_())
print(f'write only (zero_)       {us:7.1f} us  {mb / us:6.2f} TB/s')
us = timed(lambda i: bufs[i % 6].sum())
print(f'read only (torch.sum)    {us:7.1f} us  {mb / us:6.2f} TB/s')
us = timed(lambda i: out[i % 6].copy_(bufs[i % 6]))
print(f'copy (1 R + 1 W)         {us:7.1f} us  {2 * mb / us:6.2f} TB/s')
us = timed(lambda i: torch.add(bufs[i % 6], bufs[(i + 1) % 6], out=out[i % 6]))
print(f'add (2 R + 1 W)          {us:7.1f} us  {3 * mb / us:6.2f} TB/s')
us = timed(lambda i: torch.addcmul(bufs[i % 6], bufs[(i + 1) % 6], bufs[(i + 2) % 6], out=out[i % 6]))
print(f'addcmul (3 R + 1 W)      {us:7.1f} us  {4 * mb / us:6.2f} TB/s')

# the SURVEY 8(d3) unit's two planes (B = 1, N = 50 176, F = 1024 floats: 205.5 MB each), buffers rotated so that nothing is cache-resident
m = 50176 * 1024
src = [torch.randn(m, device=dev) for _ in range(6)]
dst = [torch.empty(m, device=dev) for _ in range(6)]
us = timed(lambda i: dst[i % 6].copy_(src[i % 6]), reps=60)
print(f'copy of the 8(d3) unit   {us:7.1f} us  {2 * m * 4 / 1e6 / us:6.2f} TB/s   (the unit itself: 414.4 MB of algorithmic bytes)')
