import sys, time, torch
sys.path.insert(0, "/root/repo")
from sola_amd import _lib
from sola_amd._lib import lib, ptr, current_stream, check
x = torch.zeros(64, device="cuda"); p = torch.empty(64, device="cuda"); q = torch.empty(64, device="cuda")
st = current_stream(x.device)
for _ in range(100): lib().sola_select(ptr(x), 64, 0.5, ptr(p), ptr(q), st)
torch.cuda.synchronize()
for n in (2000, 2000):
    t0 = time.perf_counter()
    for _ in range(n): lib().sola_select(ptr(x), 64, 0.5, ptr(p), ptr(q), st)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"sola_select x{n}: host {1e6*(t1-t0)/n:.2f} us per call, wall {1e6*(t2-t0)/n:.2f} us per call")
# pure ctypes overhead: a call that launches nothing
t0 = time.perf_counter()
for _ in range(20000): lib().sola_has_experiments()
print(f"ctypes no-op: {1e6*(time.perf_counter()-t0)/20000:.2f} us per call")
