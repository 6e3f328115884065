#!/usr/bin/env python3
"""What a write-only stream sustains on this GPU (GPU box): torch fill_ / zero_ of 1, 4 and 16 GB, HIP events, loaded clocks.
The fused rollout writes 9 bytes for every byte it reads from HBM, so this -- not the 8 TB/s read+write peak -- is the
physical ceiling of its output stream."""
import torch
dev = torch.device("cuda:0")
for gb in (1, 4, 16):
    x = torch.empty(gb * (1 << 30) // 4, dtype=torch.float32, device=dev)
    for name, fn in (("fill_", lambda: x.fill_(1.5)), ("zero_", lambda: x.zero_())):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ev = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); ev.append((e0, e1))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        print(f"{name} {gb:2d} GB: median {ms[5]:.3f} ms = {gb * 1.073741824 / ms[5]:.2f} TB/s, best {gb * 1.073741824 / ms[0]:.2f} TB/s", flush=True)
    del x
y = torch.empty(4 * (1 << 30) // 4, dtype=torch.float32, device=dev); z = torch.empty_like(y)
for _ in range(3): z.copy_(y)
torch.cuda.synchronize()
ev = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); z.copy_(y); e1.record(); ev.append((e0, e1))
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
print(f"copy 4 GB -> 4 GB: median {ms[5]:.3f} ms = {8 * 1.073741824 / ms[5]:.2f} TB/s (read + write)")
