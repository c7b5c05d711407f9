"""Per-shape timing of the train step's GEMMs (bf16) vs their MFMA / HBM floors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flex-dm_amd"))
import torch
from mfp.hip import ops

T, D = 32768, 256
dev = "cuda:0"
shapes = [  # name, layout, M, N, K, out dtype, extras
    ("enc dense", "fwd", T, 256, 512, torch.float32), ("qkv", "fwd", T, 768, 256, torch.bfloat16),
    ("o-proj+res", "fwd", T, 256, 256, torch.float32), ("ffn1+relu", "fwd", T, 512, 256, torch.bfloat16),
    ("ffn2+res", "fwd", T, 256, 512, torch.float32), ("heads", "fwd", T, 1384, 256, torch.float32),
    ("dh", "dgrad", T, 512, 256, torch.bfloat16), ("dy2", "dgrad", T, 256, 512, torch.bfloat16),
    ("da", "dgrad", T, 256, 256, torch.bfloat16), ("dy1", "dgrad", T, 256, 768, torch.bfloat16),
    ("dheads", "dgrad", T, 256, 1384, torch.float32),
    ("wg ffn2", "wgrad", 256, 512, T, torch.float32), ("wg ffn1", "wgrad", 512, 256, T, torch.float32),
    ("wg o", "wgrad", 256, 256, T, torch.float32), ("wg qkv", "wgrad", 768, 256, T, torch.float32),
    ("wg heads", "wgrad", 1384, 256, T, torch.float32), ("wg enc", "wgrad", 256, 512, T, torch.float32),
]
tot = 0.0
for name, layout, M, N, K, odt in shapes:
    if layout == "fwd":
        A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16(); ak, bk = True, True
    elif layout == "dgrad":
        A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16(); ak, bk = True, False
    else:
        A = torch.randn(K, M, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16(); ak, bk = False, False
    out = torch.empty(M, N, device=dev, dtype=odt)
    sk = ops.wgrad_splitk(K, M, N) if layout == "wgrad" else 1
    def run():
        ops.gemm(A, B, M, N, K, a_kmajor=ak, b_kmajor=bk, out=out, splitk=sk)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    # hipGraph replay: the Python/ctypes launch path (~15 us per call) must not be what is timed
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): run()
    gr.replay(); torch.cuda.synchronize()
    e0.record()
    gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flops = 2.0 * M * N * K
    nbytes = (M * K + N * K) * 2 + M * N * out.element_size()
    tot += us
    print("%-12s %-6s M=%6d N=%5d K=%6d sk=%2d  %7.1f us  %6.0f TF/s  (mfma floor %5.1f us, hbm floor %5.1f us @5TB/s)" % (
        name, layout, M, N, K, sk, us, flops / us / 1e6, flops / 2.5e15 * 1e6, nbytes / 5e12 * 1e6))
print("sum %.1f us" % tot)
