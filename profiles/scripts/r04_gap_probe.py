"""Is the one-off 11-36 ms delay that profiles/scripts/r04_stall_probe.py finds on one launch of the
second / third warm setup of a process (always cm2_bd_det_mask; GPU-side: polling hipStreamQuery sees
it as well as a blocking wait; no allocation, no new torch segment in that repetition) a property of
this library's call sequence or of the platform?  A torch-only process: light kernels timed one by
one (launch + poll until idle), a burst of heavy kernels in the middle (like a plan build), light
kernels again; every light kernel that takes more than 3 ms is reported with its time since the first
GPU use of the process."""
import json, sys, time
import torch
dev = torch.device("cuda", 0)
t_first = time.perf_counter()
x = torch.zeros(1 << 20, device=dev)
heavy = torch.rand(1 << 28, device=dev)                       # 1 GiB of floats
st = torch.cuda.current_stream()
torch.cuda.synchronize()
gaps = []


def light(n, tag):
    for i in range(n):
        t0 = time.perf_counter()
        x.add_(1.0)
        while not st.query():
            pass
        dt = time.perf_counter() - t0
        if dt > 3e-3:
            gaps.append({"phase": tag, "kernel_index": i, "ms": round(1e3 * dt, 2),
                         "s_since_first_gpu_use": round(t0 - t_first, 3)})


light(2000, "before the burst")
tb = time.perf_counter()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for _ in range(6):
        heavy.sort()                                           # ~ the sorts of a plan build
    torch.cuda.synchronize()
    light(3000, "after burst %d" % rep)
print(json.dumps({"seconds": round(time.perf_counter() - t_first, 2), "light_kernels_over_3ms": gaps}))
