#!/usr/bin/env python
"""Stand-alone timing of the grouped weight-gradient launch (csrc/gemm_wgg.h) at the timed shape:
the four products of a DeepSVG block, the heads, the encoder tail -- next to the per-product path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch  # noqa: E402

from mfp.hip import ops  # noqa: E402

T, D = int(os.environ.get("T", 32768)), int(os.environ.get("D", 256))
dev = "cuda"


def rnd(*shape):
    return torch.randn(*shape, device=dev).to(torch.bfloat16)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def block_jobs():
    dqkv, y1, dh, y2, d_o2, h, d_o1, a = rnd(T, 3 * D), rnd(T, D), rnd(T, 2 * D), rnd(T, D), rnd(T, D), rnd(T, 2 * D), rnd(T, D), rnd(T, D)
    f = lambda *s: torch.empty(*s, device=dev)
    return [dict(A=dqkv, B=y1, out=f(3 * D, D), M=3 * D, N=D, colsum=f(3 * D)), dict(A=dh, B=y2, out=f(2 * D, D), M=2 * D, N=D, colsum=f(2 * D)),
            dict(A=d_o2, B=h, out=f(D, 2 * D), M=D, N=2 * D), dict(A=d_o1, B=a, out=f(D, D), M=D, N=D)]


def main():
    jobs = block_jobs()
    byts = sum(T * (j["M"] + j["N"]) * 2 for j in jobs)
    flops = sum(2 * T * j["M"] * j["N"] for j in jobs)
    for sk in (None, 8, 16, 24, 32):
        us = timeit(lambda: ops.wgrad_group(jobs, T, sk))
        print("block group splitk=%s: %.1f us  (%.2f TB/s operands, %.0f TFLOP/s)" % (sk, us, byts / us / 1e6, flops / us / 1e6))
    def old():
        for j in jobs:
            ops.gemm(j["A"], j["B"], j["M"], j["N"], T, a_kmajor=False, b_kmajor=False, out=j["out"], colsum=j.get("colsum"),
                     splitk=ops.wgrad_splitk(T, j["M"], j["N"]))
    print("block, 4 x (mfp_gemm + reduce): %.1f us" % timeit(old))
    U = 1384
    hj = [dict(A=rnd(T, U), B=rnd(T, D), out=torch.empty(U, D, device=dev), M=U, N=D, colsum=torch.empty(U, device=dev))]
    for sk in (None, 8, 16, 24):
        print("heads group splitk=%s: %.1f us" % (sk, timeit(lambda: ops.wgrad_group(hj, T, sk))))
    print("heads mfp_gemm + reduce: %.1f us" % timeit(lambda: ops.gemm(hj[0]["A"], hj[0]["B"], U, D, T, a_kmajor=False, b_kmajor=False,
                                                                 out=hj[0]["out"], colsum=hj[0]["colsum"], splitk=ops.wgrad_splitk(T, U, D))))
    code = (torch.rand(T, device=dev) < 0.3).to(torch.uint8)
    dhc = rnd(T, D)
    ej = [dict(A=dhc, B=rnd(T, 512), out=torch.empty(D, 512, device=dev), M=D, N=512, rowskip=code, colsum=torch.empty(D, device=dev)),
          dict(A=dhc, B=rnd(T, 512), out=torch.empty(D, 512, device=dev), M=D, N=512, rowskip=code, colsum=torch.empty(D, device=dev)),
          dict(A=rnd(T, 344), B=dhc, out=torch.empty(344, D, device=dev), M=344, N=D)]
    for sk in (None, 8, 16):
        print("encoder tail group splitk=%s: %.1f us" % (sk, timeit(lambda: ops.wgrad_group(ej, T, sk))))


if __name__ == "__main__":
    main()
