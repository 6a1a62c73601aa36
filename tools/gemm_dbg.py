import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
def tune(**kv):
    for k, v in kv.items(): lib.sola_tune(k.encode(), int(v))
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1000, 520, 96)
rng = np.random.default_rng(1)
a = rng.standard_normal((M, K)).astype(np.float32); w = (rng.standard_normal((N, K)) * 0.03).astype(np.float32)
b = rng.standard_normal(N).astype(np.float32); r = rng.standard_normal((M, N)).astype(np.float32)
asp, wsp, bd, rd = ops.cast_sp16(torch.from_numpy(a).cuda()), ops.cast_sp16(torch.from_numpy(w).cuda(), 64.0), torch.from_numpy(b).cuda(), torch.from_numpy(r).cuda()
rsp = ops.cast_sp16(rd)
base = ops.decode_sp16(asp).double().cpu().numpy() @ (ops.decode_sp16(wsp).double().cpu().numpy() / 64.0).T + b
tune(gemm_glds_force=1)
for name, st in [("128", dict(gemm_glds=1, gemm_persist=0)), ("256", dict(gemm_glds=4, gemm_persist=0)), ("256p", dict(gemm_glds=4, gemm_persist=1))]:
    tune(**st)
    for res, rs, osp in [(None, False, False), (rd, False, False), (rsp, True, False), (None, False, True), (rsp, True, True)]:
        out = torch.full((M, N), 777.0, device="cuda")
        _lib.check(lib.sola_gemm_nt_split(_lib.ptr(asp), K, _lib.ptr(wsp), _lib.ptr(bd), _lib.ptr(res), N, 1 if rs else 0, _lib.ptr(out), N, 1 if osp else 0, M, N, K, 1 / 64.0, _lib.current_stream(out.device)), "g")
        torch.cuda.synchronize()
        got = (ops.decode_sp16(out) if osp else out).double().cpu().numpy()
        ref = base + (0 if res is None else r)
        err = np.abs(got - ref)
        bad = np.argwhere(err > 1e-4 * max(1, np.abs(ref).max()))
        print(name, "res", res is not None, "rs", rs, "osp", osp, "max err", err.max(), "bad", len(bad), bad[:3].tolist(), bad[-2:].tolist() if len(bad) else "")
