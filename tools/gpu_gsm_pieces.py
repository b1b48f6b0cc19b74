"""Where does a device-resident GSM cycle spend its time?  (dev)  Times the pieces of gsm.GrowingStringDriver._device_cycle on the GPU at
K images x D coordinates: wall time per call with a synchronisation after each (so launch + execution + any hidden host sync shows).

    python tools/gpu_gsm_pieces.py [K] [atoms]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from pdb2reaction_amd import gsm  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
d = 3 * (int(sys.argv[2]) if len(sys.argv) > 2 else 2000)
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.cumsum(torch.randn(k, d, dtype=torch.float64, generator=g), 0).to(dev)
f = torch.randn(k, d, dtype=torch.float64, generator=g).to(dev)
m = 10
hs = torch.randn(m, k * d, dtype=torch.float64, generator=g).to(dev)
hy = hs * 1.3 + 0.01 * torch.randn(m, k * d, dtype=torch.float64, generator=g).to(dev)
gr = torch.randn(k * d, dtype=torch.float64, generator=g).to(dev)
tg = torch.linspace(0, 1, k, dtype=torch.float64, device=dev)


def timeit(name, fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_enq = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"{name:28s} enqueue {t_enq * 1e3:8.3f} ms   total {t_all * 1e3:8.3f} ms per call")


seg = (x[1:] - x[:-1]).norm(dim=1)
u = torch.cat([seg.new_zeros(1), seg.cumsum(0)])
timeit("spline_derivative_t", lambda: gsm.spline_derivative_t(u, x))
timeit("tangents_t spline", lambda: gsm.tangents_t(x, "spline"))
timeit("tangents_t central", lambda: gsm.tangents_t(x, "central"))
timeit("place_t", lambda: gsm.place_t(x, tg))
timeit("lbfgs_direction_t (m=10)", lambda: gsm.lbfgs_direction_t(hs, hy, gr))
timeit("lbfgs_two_loop_t (m=10)", lambda: gsm.lbfgs_two_loop_t(hs, hy, gr))
timeit("hei_index_t", lambda: gsm.hei_index_t(f[:, 0]))
timeit("projection", lambda: (f - (f * x).sum(1, keepdim=True) * x))
timeit("cat history", lambda: torch.cat([hs, gr[None]])[-10:])
timeit("stats .cpu()", lambda: torch.cat([f[:, 0], f[:, 1]]).cpu())
a = torch.randn(k, k, dtype=torch.float64, device=dev) + 4 * torch.eye(k, dtype=torch.float64, device=dev)
timeit("linalg.solve_ex 16x16 x D", lambda: torch.linalg.solve_ex(a, x, check_errors=False))
timeit("linalg.inv_ex 16x16", lambda: torch.linalg.inv_ex(a, check_errors=False))
r = torch.triu(torch.randn(m, m, dtype=torch.float64, device=dev)) + 3 * torch.eye(m, dtype=torch.float64, device=dev)
timeit("solve_triangular 10x10", lambda: torch.linalg.solve_triangular(r, gr[:m, None], upper=True))
timeit("S @ Y.T (10 x 96k)", lambda: hs @ hy.T)
timeit("_gram chunked bmm", lambda: gsm._gram(hs, hy))
timeit("gram broadcast-sum", lambda: (hs[:, None, :] * hy[None, :, :]).sum(-1))
timeit("gemv hs @ g", lambda: hs @ gr)
timeit("gemv-T hs.T @ p", lambda: hs.T @ gr[:m])
