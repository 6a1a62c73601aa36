"""Host-side cost of staging tokens for upload: copy of a 4 MB tensor into pageable memory, into a RECYCLED page-locked buffer, and
Tensor.pin_memory() (fresh page-locked memory), by number of threads; and the host-to-device copy rate from each."""
import sys, time, threading
import torch
torch.cuda.init()
N = 1 << 20  # floats
def bench(fn, nthreads, reps=40):
    def work():
        for _ in range(reps): fn()
    ths = [threading.Thread(target=work) for _ in range(nthreads)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    return nthreads * reps * N * 4 / dt / 1e9
for nt in (1, 4, 16, 32):
    srcs = [torch.randn(N) for _ in range(nt)]
    pag = [torch.empty(N) for _ in range(nt)]
    pin = [torch.empty(N, pin_memory=True) for _ in range(nt)]
    loc = threading.local()
    idx = iter(range(10 ** 9)); lock = threading.Lock()
    def slot():
        if not hasattr(loc, "i"):
            with lock: loc.i = next(idx) % nt
        return loc.i
    r = []
    r.append(bench(lambda: pag[slot()].copy_(srcs[slot()]), nt))
    r.append(bench(lambda: pin[slot()].copy_(srcs[slot()]), nt))
    r.append(bench(lambda: srcs[slot()].pin_memory(), nt, reps=10))
    print(f"{nt:2d} threads: copy to pageable {r[0]:6.1f} GB/s   to recycled pinned {r[1]:6.1f} GB/s   pin_memory() {r[2]:6.1f} GB/s", flush=True)
d = torch.empty(N, device="cuda")
for name, h in (("pageable", torch.randn(N)), ("pinned", torch.randn(N).pin_memory())):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"H2D from {name}: {50 * N * 4 / dt / 1e9:.1f} GB/s")
