"""Practical streaming ceilings of the box: device-to-device copy, read-only sum and fill of 1.6 GB."""
import torch, json
dev = torch.device("cuda", 0)
n = 200_000_000
x = torch.rand(n, device=dev, dtype=torch.float64)
y = torch.empty_like(x)
def ev(fn, reps=9):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
t = ev(lambda: y.copy_(x)); out = {"copy_1.6GB_read_plus_1.6GB_write_ms": round(t, 4), "copy_TBps": round(2 * 8 * n / t / 1e9, 3)}
t = ev(lambda: x.sum()); out["sum_ms"] = round(t, 4); out["read_TBps"] = round(8 * n / t / 1e9, 3)
t = ev(lambda: y.fill_(1.0)); out["fill_ms"] = round(t, 4); out["write_TBps"] = round(8 * n / t / 1e9, 3)
t = ev(lambda: torch.add(x, 1.0, out=y)); out["add_ms"] = round(t, 4); out["add_TBps"] = round(2 * 8 * n / t / 1e9, 3)
print(json.dumps(out))
