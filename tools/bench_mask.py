"""Stand-alone timing (hipGraph replay) of the fused masking kernel at the Crello c2 shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp.data.spec import make_input_columns, synthetic_batch
from mfp.models.mfp import MFP

B, S = int(os.environ.get("B", 256)), int(os.environ.get("S", 128))
DEV = "cuda:0"
ic = make_input_columns("crello")
batch = synthetic_batch(ic, B, S, seed=3, ragged=True, device=DEV)
model = MFP(ic, num_blocks=1, latent_dim=256, dropout=0.0, l2=1e-2, dtype="bf16", device=DEV, seed=11, masking_method="random")
tasks = torch.zeros(B, dtype=torch.int32, device=DEV)
step = torch.zeros(1, dtype=torch.int32, device=DEV)
ctx = model.model.make_ctx(batch, True)
fn = lambda: model._masker(batch, tasks, ctx.nvalid, B, S, step)
for _ in range(3): fn()
torch.cuda.synchronize()
reps = 20
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(reps): fn()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
nb = B * S * 2 * 512 * 6
print("mask %-30s %.1f us  (%.0f MB numerical rows -> %.2f TB/s)" % (os.environ.get("MFP_HIP_LIB", "default")[-24:], us, nb / 1e6, nb / us / 1e6))
