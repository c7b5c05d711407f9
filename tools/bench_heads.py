"""Stand-alone timing (hipGraph replay) of the one-launch heads + loss + input-gradient kernel against the four launches it
replaces, Crello heads, T = 32768 (c2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp.hip import ops
from mfp.data.spec import make_input_columns, synthetic_batch
from mfp.models.params import ModelLayout
B, S, D, dev = int(os.environ.get("B", 256)), 128, 256, "cuda"
T = B * S
ic = make_input_columns("crello")
lay = ModelLayout(ic, 256, 1)
U = lay.Upad
batch = synthetic_batch(ic, B, S, seed=4, ragged=True)
g = torch.Generator().manual_seed(1)
x = torch.randn(T, D, generator=g).to(dev, torch.bfloat16)
W = (torch.randn(U, D, generator=g) * 0.05).to(dev, torch.bfloat16)
bias = (torch.randn(U, generator=g) * 0.1).to(dev)
descr, keep = [], []
types = batch["type"].to(dev)
for k in lay.head_order:
    c = ic[k]
    o, n = lay.head_cols[k]
    tgt = batch[k].to(dev).contiguous()
    msk = (torch.rand(B, S, generator=g) < float(os.environ.get("MASK", 0.15))).to(torch.uint8).to(dev).contiguous()
    keep += [tgt, msk]
    d = dict(col_off=o, n_feat=c["shape"][-1] if c["type"] == "categorical" else 1,
             n_class=c["input_dim"] if c["type"] == "categorical" else c["shape"][-1],
             is_numerical=c["type"] != "categorical", target=tgt, mask=msk)
    if "loss_condition" in c:
        d.update(cond_idx=types, cond_stride=1, cond_bits=sum(1 << i for i, f in enumerate(c["loss_condition"]["mask"]) if f))
    descr.append(d)
nvalid = (batch["length"].reshape(-1) + 1).to(torch.int32).to(dev)
step = torch.zeros(1, dtype=torch.int32, device=dev)
ldw = (U + 127) // 128 * 128
Wt = torch.zeros(D, ldw, dtype=torch.bfloat16, device=dev)
Wt[:, :U] = W.t()
dl = torch.zeros(T, U, dtype=torch.bfloat16, device=dev)


def timeit(name, fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    print("%-34s %7.1f us" % (name, e0.elapsed_time(e1) * 1e3 / reps))


timeit("heads_loss (1 launch, no logits)", lambda: ops.heads_loss_fused(x, W, bias, descr, nvalid, B, S, dlogits=dl, want_logits=False, drop=(0.1, 5, 8, step)))
timeit("heads_loss (1 launch, logits)", lambda: ops.heads_loss_fused(x, W, bias, descr, nvalid, B, S, dlogits=dl, want_logits=True, drop=(0.1, 5, 8, step)))
def four():
    lg = ops.gemm(x, W, T, U, D, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, bias=bias)
    ops.loss_fwd_bwd(lg, descr, nvalid, B, S, torch.bfloat16, dlogits=dl)
    return ops.dgrad_rows(dl, Wt, U, drop=(0.1, 5, 8, step))
timeit("gemm + ce + mse + dgrad_rows (4)", four)
