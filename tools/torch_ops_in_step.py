"""Which aten ops launch kernels inside one eager train step (glue hunting)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
from mfp.data.spec import make_input_columns, synthetic_batch
from mfp.models.mfp import MFP

B, S = 256, 128
ic = make_input_columns("crello")
batch = synthetic_batch(ic, B, S, seed=0, ragged=True, device="cuda:0")
model = MFP(ic, num_blocks=4, latent_dim=256, dropout=0.1, l2=1e-2, dtype="bf16", device="cuda:0", seed=0, masking_method="random")
model.compile(learning_rate=1e-4)
for _ in range(3):
    model.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    if os.environ.get("CAPTURE"):
        model.capture_train_step(batch, warmup=0)
    else:
        model.train_step(batch)
    torch.cuda.synchronize()
for e in prof.key_averages(group_by_stack_n=6):
    if e.key.startswith("aten::") and (e.self_device_time_total > 0 or (os.environ.get("CAPTURE") and ("fill" in e.key or "zero" in e.key or "ones" in e.key))):
        st = [x for x in (e.stack or [])][:6]
        print("%-28s n=%d %6.1f us  %s" % (e.key, e.count, e.self_device_time_total, " <- ".join(x.split("/")[-1][:60] for x in st)))
