import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch, numpy as np
from mfp.hip import ops
DEV = "cuda"
def run(T, N, relu):
    D = 512
    g = torch.Generator().manual_seed(500 + N + T)
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(T, D) * (0.5 + torch.rand(T, 1, generator=g)) + 0.3 * rn(T, 1)
    gam, bet = 1.0 + 0.2 * rn(D), 0.1 * rn(D)
    W, b = (rn(N, D) * 0.05).bfloat16().float(), rn(N) * 0.1
    for rep in range(3):
        out, y, mean, rstd = ops.ln_dense_d512(x.to(DEV), gam.to(DEV), bet.to(DEV), W.to(DEV, torch.bfloat16), b.to(DEV), N, relu=relu)
        torch.cuda.synchronize()
        want = y.float() @ W.to(DEV).t() + b.to(DEV)
        if relu: want = torch.relu(want)
        bad = ((out.float() - want).abs() > 0.05 + 0.02 * want.abs())
        nb = int(bad.sum())
        print("T=%d N=%d relu=%d rep=%d bad=%d" % (T, N, relu, rep, nb))
        if nb:
            idx = bad.nonzero().cpu().numpy()
            rows, cols = idx[:, 0], idx[:, 1]
            print("  tiles:", np.unique(rows // 128)[:20], "n tiles", len(np.unique(rows // 128)))
            print("  rows%128:", np.unique(rows % 128)[:40])
            print("  col groups (64):", np.unique(cols // 64))
            print("  cols%64:", np.unique(cols % 64))
            r0, c0 = rows[0], cols[0]
            print("  sample got/want", out[r0, c0 - c0 % 16:c0 - c0 % 16 + 16].float().cpu().numpy(), want[r0, c0 - c0 % 16:c0 - c0 % 16 + 16].cpu().numpy())
run(1000, 1024, True)
run(1000, 1024, False)
run(1024, 1024, True)
run(16384, 1536, False)
run(16384, 1024, True)
run(2048, 1536, False)
