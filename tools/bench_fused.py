"""Stand-alone timing (hipGraph replay) of the activation-stationary block kernels at T tokens (default 32768).
MFP_FUSED_HALF=0/1 selects the 128-row / half-size workgroups."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp.hip import ops
T, D = int(os.environ.get("T", 32768)), 256
dev = "cuda"
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)


def timeit(name, fn, nbytes, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("%-22s T=%d half=%s  %6.1f us  %5.2f TB/s" % (name, T, os.environ.get("MFP_FUSED_HALF", "auto"), us, nbytes / us / 1e6))


which = os.environ.get("WHICH", "dgrad768,dgrad256,qkv,mlp,mlpbwd,enc,rows").split(",")
if "dgrad768" in which:
    dqkv, wt = rnd(T, 768), rnd(256, 768)
    timeit("dgrad_qkv<768>", lambda: ops.dgrad_qkv(dqkv, wt), T * (768 + 256) * 2)
if "dgrad256" in which:
    dy, wt2 = rnd(T, 256), rnd(256, 256)
    timeit("dgrad_qkv<256>", lambda: ops.dgrad_d256(dy, wt2), T * 512 * 2)
x = torch.randn(T, D, device=dev)
gam, bet = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
if "qkv" in which:
    W, b = rnd(768, 256), torch.randn(768, device=dev)
    timeit("qkv_fused", lambda: ops.qkv_fused_fwd(x, gam, bet, W, b), T * (256 * 4 + 256 * 2 + 768 * 2))
if "mlp" in which:
    W1, b1, W2, b2 = rnd(512, 256), torch.randn(512, device=dev), rnd(256, 512), torch.randn(256, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    timeit("mlp_fused", lambda: ops.mlp_fused_fwd(x, gam, bet, W1, b1, W2, b2, (0.1, 5, 3), step), T * (256 * 4 * 3 + 256 * 2 + 512 * 2))
if "mlpbwd" in which:
    d_o2, h, W2t, W1t = rnd(T, 256), rnd(T, 512), rnd(512, 256), rnd(256, 512)
    timeit("mlp_bwd", lambda: ops.mlp_fused_bwd(d_o2, h, W2t, W1t), T * (256 * 2 * 2 + 512 * 2 * 2))
if "attnblock" in which:
    B = T // 128
    Wq, bq = rnd(768, 256), torch.randn(768, device=dev)
    Wo, bo = rnd(256, 256), torch.randn(256, device=dev)
    nvalid = torch.full((B,), 128, dtype=torch.int32, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    timeit("attn_block_fwd", lambda: ops.attn_block_fwd(x, gam, bet, Wq, bq, Wo, bo, nvalid, B, 128, 8, (0.1, 5, 3), step), T * (256 * 4 * 3 + 256 * 2 * 2 + 768 * 2))
    def three():
        qkv, y1, m, r = ops.qkv_fused_fwd(x, gam, bet, Wq, bq)
        a, lse = ops.attention_fwd(qkv, nvalid, B, 128, 8)
        return ops.gemm(a, Wo, T, 256, 256, a_kmajor=True, b_kmajor=True, bias=bo, residual=x, dropout=(0.1, 5, 3), step_ptr=step, out_dtype=torch.float32)
    timeit("qkv+attn+oproj (3)", three, T * (256 * 4 * 3 + 256 * 2 * 4 + 768 * 2 * 2))
if "block" in which:
    B = T // 128
    Wq, bq = rnd(768, 256), torch.randn(768, device=dev)
    Wo, bo = rnd(256, 256), torch.randn(256, device=dev)
    W1, b1, W2, b2 = rnd(512, 256), torch.randn(512, device=dev), rnd(256, 512), torch.randn(256, device=dev)
    g2, be2 = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
    nvalid = torch.full((B,), 128, dtype=torch.int32, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    timeit("block_fwd (1 launch)", lambda: ops.block_fwd(x, gam, bet, Wq, bq, Wo, bo, nvalid, g2, be2, W1, b1, W2, b2, B, 128, 8, 0.1, 5, 3, 4, step),
           T * (256 * 4 * 5 + 256 * 2 * 3 + 768 * 2 + 512 * 2))
    timeit("block_fwd_xhat", lambda: ops.block_fwd(x, gam, bet, Wq, bq, Wo, bo, nvalid, g2, be2, W1, b1, W2, b2, B, 128, 8, 0.1, 5, 3, 4, step, xhat_stash=True),
           T * (256 * 4 * 5 + 256 * 2 * 3 + 768 * 2 + 512 * 2))
    # two four-wave workgroups per document (fewer documents than CUs: T=16384 is config c4's per-GPU share)
    for wv in (4, 8):
        timeit("block_fwd_xhat_half (%d waves)" % wv, lambda: ops.block_fwd(x, gam, bet, Wq, bq, Wo, bo, nvalid, g2, be2, W1, b1, W2, b2, B, 128, 8, 0.1, 5, 3, 4, step, xhat_stash=True, half_tiles=wv),
               T * (256 * 4 * 5 + 256 * 2 * 3 + 768 * 2 + 512 * 2))
    # the inference form (nothing saved): what the in-CU pipeline does when only x1 / x2 cross HBM
    timeit("block_infer (1 launch)", lambda: ops.block_infer(x, gam, bet, Wq, bq, Wo, bo, nvalid, g2, be2, W1, b1, W2, b2, B, 128, 8),
           T * 256 * 4 * 5)
    def two():
        x1 = ops.attn_block_fwd(x, gam, bet, Wq, bq, Wo, bo, nvalid, B, 128, 8, (0.1, 5, 3), step)[0]
        return ops.mlp_fused_fwd(x1, g2, be2, W1, b1, W2, b2, (0.1, 5, 4), step)
    timeit("attn_block + mlp_fused", two, T * (256 * 4 * 6 + 256 * 2 * 3 + 768 * 2 + 512 * 2))
if "attnbwd" in which:
    B = T // 128
    qkv = (torch.randn(T, 768, device=dev) * 0.7).to(torch.bfloat16)
    nvalid = torch.full((B,), 128, dtype=torch.int32, device=dev)
    a, lse = ops.attention_fwd(qkv, nvalid, B, 128, 8)
    d_o1, Wot, Wqt = rnd(T, 256), rnd(256, 256), rnd(256, 768)
    timeit("attn_block_bwd (1)", lambda: ops.attn_block_bwd(d_o1, Wot, qkv, a, lse, nvalid, Wqt, B, 128, 8), T * (256 * 2 * 3 + 768 * 2 * 2))
    def three_b():
        da = ops.dgrad_d256(d_o1, Wot)
        dq = ops.attention_bwd(qkv, nvalid, a, da, lse, B, 128, 8)
        return ops.dgrad_qkv(dq, Wqt)
    timeit("dgrad+attn_bwd+dgrad (3)", three_b, T * (256 * 2 * 5 + 768 * 2 * 3))
