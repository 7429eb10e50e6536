"""What the HBM takes in WRITES on this box: torch's fill of 1 / 4 / 10 GB (a plain streaming store kernel), GB/s by CUDA events --
the practical ceiling under the G-buffer stores of the bench line (10.15 GB per 128-frame launch, profiles/r04_k_primary_pmc.json)."""
import torch
dev = torch.device("cuda:0")
for gb in (1, 4, 10):
    x = torch.empty(gb * (1 << 30), dtype=torch.uint8, device=dev)
    for _ in range(3): x.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): x.zero_()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"WRITEBW fill {gb} GiB: {ms:.3f} ms, {gb * (1 << 30) / ms / 1e6:.0f} GB/s", flush=True)
    y = torch.empty_like(x)
    for _ in range(2): y.copy_(x)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"WRITEBW copy {gb} GiB: {ms:.3f} ms, {2 * gb * (1 << 30) / ms / 1e6:.0f} GB/s (read + write)", flush=True)
    del x, y
