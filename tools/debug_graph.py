import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flex-dm_amd"))
import torch
from mfp.data.spec import make_input_columns, synthetic_batch
from mfp.models.mfp import MFP
dtype = sys.argv[1]; B = int(sys.argv[2]); drop = float(sys.argv[3]); mode = sys.argv[4]
dev = "cuda:0"
ic = make_input_columns("crello")
batch = synthetic_batch(ic, B, 128, seed=0, ragged=False, device=dev)
model = MFP(ic, num_blocks=4, latent_dim=256, dropout=drop, l2=1e-2, dtype=dtype, device=dev)
model.compile(learning_rate=1e-4)
if mode == "graph":
    model.capture_train_step(batch, warmup=2)
for step in range(6):
    sums = model.train_step(batch)
    torch.cuda.synchronize()
    st = model.model.store
    print(mode, dtype, drop, step, "loss %.4g" % float(sums[:, 0].sum()), "sums0", [round(x, 1) for x in sums[:, 0].tolist()],
          "gmax %.3g wmax %.3g statsmax %.3g" % (st.g.abs().max().item(), st.w.abs().max().item(), model.optimizer.stats.abs().max().item()))
