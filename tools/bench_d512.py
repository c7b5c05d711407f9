"""Stand-alone timing (hipGraph replay) of the d_model-512 Dense kernels (csrc/block_d512.hip) at c5's token count against the
launches they replace (ln_fwd + weight-stationary / LDS-tiled mfp_gemm)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp.hip import ops
T, D = int(os.environ.get("T", 16384)), 512
dev = "cuda"
bf = torch.bfloat16
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(bf)


def timeit(name, fn, nbytes, flops, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("%-34s T=%d  %6.1f us  %5.2f TB/s  %6.1f TFLOP/s" % (name, T, us, nbytes / us / 1e6, flops / us / 1e6), flush=True)


x = torch.randn(T, D, device=dev)
gam, bet = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
step = torch.zeros(1, dtype=torch.int32, device=dev)
for N, relu, what in ((1536, False, "LN1 + QKV"), (1024, True, "LN2 + FFN1")):
    W, b = rnd(N, D), torch.randn(N, device=dev)
    nb = T * (D * 4 + D * 2 + N * 2)
    timeit("as512 %s" % what, lambda: ops.ln_dense_d512(x, gam, bet, W, b, N, relu=relu), nb, 2 * T * D * N)
    def old():
        y, m, r = ops.layernorm_fwd(x, gam, bet, bf)
        return ops.gemm(y, W, T, N, D, a_kmajor=True, b_kmajor=True, bias=b, relu=relu, out_dtype=bf)
    timeit("ln_fwd + mfp_gemm %s" % what, old, nb + T * D * 2, 2 * T * D * N)
A5, W2t, h = rnd(T, D), rnd(1024, D), torch.relu(torch.randn(T, 1024, device=dev)).to(bf)
timeit("as512 dh (mask)", lambda: ops.dense_relumask_d512(A5, W2t, h), T * (D * 2 + 1024 * 4), 2 * T * D * 1024)
timeit("mfp_gemm dh (relu_bwd)", lambda: ops.gemm(A5, W2t, T, 1024, D, a_kmajor=True, b_kmajor=True, out_dtype=bf, relu_bwd_aux=h),
       T * (D * 2 + 1024 * 4), 2 * T * D * 1024)
res = torch.randn(T, D, device=dev)
bo = torch.randn(D, device=dev)
for K, what in ((512, "O-proj"), (1024, "FFN2")):
    A, W = rnd(T, K), rnd(D, K)
    nb = T * (K * 2 + D * 8)
    timeit("os512 %s (res, dropout)" % what, lambda: ops.dense_n512_res(A, W, bo, res, (0.1, 5, 3), step), nb, 2 * T * D * K)
    timeit("mfp_gemm %s" % what, lambda: ops.gemm(A, W, T, D, K, a_kmajor=True, b_kmajor=True, bias=bo, residual=res, dropout=(0.1, 5, 3),
                                                 step_ptr=step, out_dtype=torch.float32), nb, 2 * T * D * K)
for K, what in ((512, "da"), (1024, "dy2"), (1536, "dy1")):
    A, W = rnd(T, K), rnd(D, K)
    nb = T * (K * 2 + D * 2)
    timeit("os512 %s (bf16)" % what, lambda: ops.dense_n512(A, W), nb, 2 * T * D * K)
    timeit("mfp_gemm %s" % what, lambda: ops.gemm(A, W, T, D, K, a_kmajor=True, b_kmajor=True, out_dtype=bf), nb, 2 * T * D * K)
