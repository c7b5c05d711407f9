"""Stand-alone timing (hipGraph replay) of the fused masking kernel at the c2 shape (256 documents x 128 positions, ragged lengths,
bf16 numerical rows): MFP_MASK_WAVE=1 selects the wave-per-token form; MFP_HIP_LIB an ablation build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp.data.spec import make_input_columns, synthetic_batch
from mfp.models.mfp import MFP
B, S, dev = int(os.environ.get("B", 256)), int(os.environ.get("S", 128)), "cuda"
ic = make_input_columns("crello")
batches = [synthetic_batch(ic, B, S, seed=4 + i, ragged=True, device=dev) for i in range(4)]      # 4 x 136 MB: beyond the infinity cache
model = MFP(ic, num_blocks=1, latent_dim=256, dropout=0.0, l2=1e-2, dtype="bf16", device=dev, seed=13, masking_method=os.environ.get("METHOD", "random"))
tasks = torch.zeros(B, dtype=torch.int32, device=dev)
ctxs = [model.model.make_ctx(b, True) for b in batches]
def step():
    for b, c in zip(batches, ctxs):
        model._masker(b, tasks, c.nvalid, B, S, None)
for _ in range(3): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(5): step()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print("mask_tokens  B=%d S=%d  wave-form=%s lib=%s  %6.1f us per launch" % (B, S, os.environ.get("MFP_MASK_WAVE", "0"), os.path.basename(os.environ.get("MFP_HIP_LIB", "default")), e0.elapsed_time(e1) * 1e3 / 20))
