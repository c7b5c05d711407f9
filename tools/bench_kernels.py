"""Stand-alone timing (hipGraph replay, no launch path) of the non-GEMM kernels of the train step at
the Crello c2 shape, next to the bytes each must move."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flex-dm_amd"))
import torch
from mfp.hip import ops

B, S, D, H = 256, 128, 256, 8
T = B * S
dev = "cuda"
bf = torch.bfloat16


def timeit(name, fn, nbytes, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("%-28s %7.1f us   %6.1f MB  -> %5.2f TB/s  (floor @6 TB/s %5.1f us)" % (name, us, nbytes / 1e6, nbytes / us / 1e6, nbytes / 6e6))


x = torch.randn(T, D, device=dev)
gamma, beta = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, bf)
timeit("ln_fwd (f32 -> bf16)", lambda: ops.layernorm_fwd(x, gamma, beta, bf), T * D * 6)
dy = torch.randn(T, D, device=dev).to(bf)
dres = torch.randn(T, D, device=dev)
dg, db, cs = torch.empty(D, device=dev), torch.empty(D, device=dev), torch.empty(D, device=dev)
dx = torch.empty(T, D, device=dev)
step = torch.zeros(1, dtype=torch.int32, device=dev)
timeit("ln_bwd (+dres)", lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres, dg, db, dx=dx), T * D * 14)
timeit("ln_bwd (+dres, +dropout out)", lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres, dg, db, dx=dx,
                                                               drop=(cs, 0.1, 5, 3, step)), T * D * 16)
qkv = torch.randn(T, 3 * D, device=dev).to(bf)
nvalid = torch.randint(S // 2, S + 1, (B,), dtype=torch.int32, device=dev)
out, lse = ops.attention_fwd(qkv, nvalid, B, S, H)
timeit("attn_fwd", lambda: ops.attention_fwd(qkv, nvalid, B, S, H), T * D * 8)
dout = torch.randn(T, D, device=dev).to(bf)
timeit("attn_bwd", lambda: ops.attention_bwd(qkv, nvalid, out, dout, lse, B, S, H), T * D * 16)
timeit("dropout_bwd", lambda: ops.dropout_bwd(dx, bf, cs, 0.1, 5, 3, step), T * D * 6)
