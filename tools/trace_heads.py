"""Phase timing of heads_loss_kernel from the HL_ABL=9 build (tools/abl/build_abl.sh heads_loss HL_ABL 9):
MFP_HIP_LIB=tools/abl/libmfp_heads_loss_9.so python tools/trace_heads.py"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
os.environ.setdefault("B", "256")
import torch
ns = runpy.run_path(os.path.join(ROOT, "tools", "bench_heads.py"), run_name="setup") if False else None
from mfp.hip import ops
exec(open(os.path.join(ROOT, "tools", "bench_heads.py")).read().split("def timeit")[0])
for _ in range(3):
    part, dl_, logits, dx, dxd = ops.heads_loss_fused(x, W, bias, descr, nvalid, B, S, dlogits=dl, want_logits=True, drop=(0.1, 5, 8, step))
torch.cuda.synchronize()
nwg = T // 128
tr = logits.view(torch.int64).view(-1)[:nwg * 8 * 256].view(nwg, 8, 256).cpu().double()
t0 = tr[:, :, 254].min(dim=1, keepdim=True).values
clk = float(os.environ.get("CLK_MHZ", 2080.0))
nch = int(os.environ.get("NCH", 23))
def at(i): return ((tr[:, 0, i] - t0[:, 0]).mean().item()) / clk
print("start -> first chunk landed: %.2f us" % at(0))
prev = at(0)
for c in range(nch):
    a, b, d, e = at(4 * c), at(4 * c + 1), at(4 * c + 2), at(4 * c + 3)
    print("chunk %2d: wait+barrier %5.2f  logits tile %5.2f  loss -> dl image %5.2f  dlogits out + dx product %5.2f   (ends %6.2f us)" % (c, a - prev, b - a, d - b, e - d, e))
    prev = e
print("dx out: %.2f, end %.2f us" % (at(255) - at(253), at(255)))
for c in range(8):
    b = 128 + 8 * c
    v = [at(4 * c + 1)] + [at(b + i) for i in range(6)] + [at(4 * c + 2)]
    print("chunk %d detail: tile+compact %.2f | B2 %.2f | walk %.2f | B3 %.2f | convert %.2f | tload %.2f | B4 %.2f" % ((c,) + tuple(v[i + 1] - v[i] for i in range(7))))
