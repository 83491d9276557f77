"""Calibration (GPU box): what a plain device copy of the headline kernel's byte volume achieves on this GPU --
the practical ceiling the roofline fraction should be read against (the guide quotes ~6.3 TB/s for large copies)."""
import torch

for mb in (27, 54, 108, 432, 1728):
    n = mb * 1024 * 1024 // 4
    src = torch.empty(n, dtype=torch.int32, device="cuda").random_()
    dst = torch.empty_like(src)
    for _ in range(5):
        dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"copy {mb:5d} MB -> {2 * mb} MB of traffic: {us:8.2f} us  = {2 * mb * 1.048576 / us:6.2f} TB/s (read + write)")
    # write-only: fill
    e0.record()
    for _ in range(reps):
        dst.fill_(7)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"fill {mb:5d} MB: {us:8.2f} us  = {mb * 1.048576 / us:6.2f} TB/s (write only)")
