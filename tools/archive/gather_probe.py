"""Dev-only probe: how many random 4-byte reads per second does the memory system deliver, by table size?
(independent of our kernels: torch index_select)."""
import torch, time
dev = "cuda"
n_idx = 100_000_000
for mb in (8, 64, 256, 1024, 4096, 16384):
    n = mb * 1024 * 1024 // 4
    table = torch.arange(n, dtype=torch.int32, device=dev)
    idx = torch.randint(0, n, (n_idx,), device=dev, dtype=torch.int64)
    out = table[idx]; torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        out = table[idx]
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"table {mb:6d} MB: {ms:7.3f} ms for {n_idx/1e6:.0f}M random 4-B reads -> {n_idx/ms/1e6:7.1f} G reads/s")
    del table, idx, out
