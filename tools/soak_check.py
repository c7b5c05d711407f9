"""Soak run of the captured train step on the shapes whose kernels synchronise across workgroups or run on half tiles:
c5-like (d_model 512: mfp_dense_n512_lnb's flag exchange, 16 launches per step), c4-like (128 documents: half tiles) and
--seq 64 at 256 documents.  Prints loss / score every 500 steps; asserts the loss falls and stays finite.
STEPS=3000 python tools/soak_check.py   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp.data.spec import make_input_columns, synthetic_batch
from mfp.models.mfp import MFP
ic = make_input_columns("crello")
dev = "cuda:0"
steps = int(os.environ.get("STEPS", 3000))
only = os.environ.get("ONLY", "")
for name, D, L, B, S in (("c5-like", 512, 8, 64, 256), ("c4-like", 256, 4, 128, 128), ("seq64 x 256", 256, 4, 256, 64)):
    if only and only not in name:
        continue
    model = MFP(ic, num_blocks=L, latent_dim=D, dropout=0.1, l2=1e-2, masking_method="random", dtype="bf16", device=dev, seed=0)
    model.compile(learning_rate=1e-4, clipnorm=1.0)
    batches = [synthetic_batch(ic, B, S, seed=s, ragged=True, device=dev) for s in range(2)]
    model.capture_train_step(batches[0], warmup=2)
    t0 = time.time()
    first = last = None
    for it in range(steps + 1):
        sums = model.train_step(batches[it % 2])
        if it % max(500, steps // 6) == 0:
            m = model.metrics_dict(sums)
            print(name, it, "loss %.3f total_score %.4f  (%.1f s)" % (m["loss"], m["total_score"], time.time() - t0), flush=True)
            first = m["loss"] if first is None else first
            last = m["loss"]
    assert last == last and last < first, (name, first, last)
    del model
    torch.cuda.empty_cache()
print("soak ok")
