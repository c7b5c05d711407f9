import sys
sys.path.insert(0, "flex-dm_amd")
import torch
from mfp.data.spec import make_input_columns, synthetic_batch
from mfp.models.mfp import MFP
ic = make_input_columns("crello")
dev = "cuda:0"
model = MFP(ic, num_blocks=4, latent_dim=256, dropout=0.1, l2=1e-2, masking_method="random", dtype="bf16", device=dev, seed=0)
model.compile(learning_rate=1e-3, clipnorm=1.0)
batches = [synthetic_batch(ic, 64, 128, seed=s, ragged=True, device=dev) for s in range(4)]   # S = 128: the document-tile kernels
model.capture_train_step(batches[0], warmup=1)
for it in range(401):
    sums = model.train_step(batches[it % 4])
    if it % 100 == 0:
        m = model.metrics_dict(sums)
        print(it, "loss %.3f total_score %.4f" % (m["loss"], m["total_score"]), flush=True)
assert m["loss"] == m["loss"]
