#!/usr/bin/env python3
"""Probe: does a consumer kernel that reads what the previous kernel just wrote run faster when the tensor fits the 256 MiB
Infinity Cache?  Ring of three tensors a -> b -> c -> a through torch.add(x, 1, out=y) (reads one tensor, writes one):
every launch reads the tensor the previous launch wrote.  Prints effective GB/s (read + write bytes) per size."""
import torch
dev = torch.device('cuda')
for mb in (16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024):
    n = mb * (1 << 20) // 4
    ring = [torch.zeros(n, device=dev) for _ in range(3)]
    iters = max(12, 6144 // mb)
    iters -= iters % 3
    for i in range(6):
        torch.add(ring[i % 3], 1.0, out=ring[(i + 1) % 3])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        torch.add(ring[i % 3], 1.0, out=ring[(i + 1) % 3])
    e.record()
    torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / iters
    print(f'{mb:5d} MB tensors: {us:8.1f} us per launch, {2 * mb * 1.048576 / us * 1e3:8.1f} GB/s (read-after-write ring of 3)')
    del ring
# cold reference: ring large enough that nothing is resident
n = 1024 * (1 << 20) // 4
