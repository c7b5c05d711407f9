"""Stand-alone timing (hipGraph replay) of the loss kernels (CE + MSE) at the Crello c2 shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp.data.spec import make_input_columns, synthetic_batch
from mfp.hip import ops
from mfp.models.metrics import build_loss_keys
from mfp.models.mfp import MFP
B, S = int(os.environ.get("B", 256)), 128
DEV = "cuda:0"
ic = make_input_columns("crello")
batch = synthetic_batch(ic, B, S, seed=3, ragged=True, device=DEV)
model = MFP(ic, num_blocks=1, latent_dim=256, dropout=0.0, l2=1e-2, dtype="bf16", device=DEV, seed=11, masking_method="random")
tasks = torch.zeros(B, dtype=torch.int32, device=DEV)
step = torch.zeros(1, dtype=torch.int32, device=DEV)
ctx = model.model.make_ctx(batch, True)
idx_all, codes, xs, masks = model._masker(batch, tasks, ctx.nvalid, B, S, step)
keys = build_loss_keys(ic, model.model.layout.head_cols, batch, masks)
U = model.model.layout.Upad
logits = torch.randn(B * S, U, device=DEV)
dl = torch.zeros(B * S, U, dtype=torch.bfloat16, device=DEV)
fn = lambda: ops.loss_fwd_bwd(logits, keys, ctx.nvalid, B, S, torch.bfloat16, dlogits=dl)
for _ in range(3): fn()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10): fn()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print("loss kernels (zero + ce + mse) %-26s %.1f us per call" % (os.environ.get("MFP_HIP_LIB", "default")[-24:], e0.elapsed_time(e1) * 100))
