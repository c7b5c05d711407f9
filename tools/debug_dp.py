import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flex-dm_amd"))
import torch, torch.distributed as dist
from mfp import dp
from mfp.data.spec import make_input_columns, synthetic_batch
from mfp.models.mfp import MFP
rank = int(os.environ["RANK"]); torch.cuda.set_device(0); dev = "cuda:0"
world = dp.init_from_env()
mode = sys.argv[1]
torch.manual_seed(1234 + rank)
ic = make_input_columns("crello")
batch = synthetic_batch(ic, 32, 128, seed=rank, device=dev)
model = MFP(ic, num_blocks=4, latent_dim=256, dropout=0.1, l2=1e-2, dtype="bf16", device=dev, seed=0)
model.compile(learning_rate=1e-4)
dp.broadcast_parameters(model.model.store.w); model.model.store.refresh_shadow()
def report(tag):
    st = model.model.store
    for name, buf in (("w", st.w), ("g", st.g), ("m", model.optimizer.m)):
        lo, hi = buf.clone(), buf.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        diff = (hi - lo).abs()
        if rank == 0:
            bad = [(n, float(diff[s.offset:s.offset + s.size].max())) for n, s in st.layout.segments.items() if diff[s.offset:s.offset + s.size].max() > 0]
            print(tag, name, "max diff %.3e" % diff.max().item(), "finite", bool(torch.isfinite(buf).all()), "n_bad_segments", len(bad), bad[:4])
if mode == "graph":
    model.capture_train_step(batch, warmup=2)
    report("after capture")
for i in range(3):
    model.train_step(batch if mode != "graph" else model.static_batch)
    torch.cuda.synchronize()
    report("step %d" % i)
dist.barrier(); dist.destroy_process_group()
