"""What a plain device-to-device copy reaches on this GPU (read + write bytes per second): the roof of any kernel
whose output is as large as its input, e.g. stage 1's raw blocks.  usage: python tools/copy_roof.py [GiB]"""
import sys
import torch
g = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = int(g * 2**30)
x = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
y = torch.empty_like(x)
for _ in range(2):
    y.copy_(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(5):
    e0.record(); y.copy_(x); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(f"copy of {g:.1f} GiB: {best:.3f} ms = {2 * n / best / 1e6:.0f} GB/s read+write ({n / best / 1e6:.0f} GB/s each way)")
