import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flex-dm_amd"))
import torch
from mfp.data.spec import make_input_columns, synthetic_batch
from mfp.models.mfp import MFP, preprocess_for_train
from mfp.models.metrics import build_loss_keys
from mfp.hip import ops, functions

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
ic = make_input_columns("crello")
batch = synthetic_batch(ic, B, 128, seed=0, ragged=False, device=dev)
model = MFP(ic, num_blocks=4, latent_dim=256, dropout=0.1, l2=1e-2, dtype=dtype, device=dev)
model.compile(learning_rate=1e-4)
def chk(name, t):
    t = t.float()
    print("%-28s shape=%s finite=%s absmax=%.4g mean=%.4g" % (name, tuple(t.shape), bool(torch.isfinite(t).all()), t.abs().max().item(), t.mean().item()))
for step in range(3):
    tasks = model.sample_tasks(B)
    targets, modified, masks = preprocess_for_train(batch, model.input_columns, tasks, active_tasks=model._active_tasks)
    for k, v in modified.items():
        if v.is_floating_point(): chk("modified " + k, v)
    ctx = model.model.make_ctx(modified, True)
    h, mask = model.model.encoder(modified, ctx)
    chk("encoder h", h)
    x = h
    for i, layer in enumerate(model.model.blocks.seq2seq.values()):
        x = layer(x, ctx)
        chk("block %d" % i, x)
    keys = build_loss_keys(ic, model.model.layout.head_cols, targets, masks)
    loss, sums, logits = functions.DecoderLossFn.apply(x.reshape(-1, 256), ctx, keys)
    chk("logits", logits)
    print("sums", sums.cpu())
    loss.backward()
    chk("grad g", model.model.store.g)
    for name, seg in model.model.store.layout.segments.items():
        gseg = model.model.store.g[seg.offset:seg.offset+seg.size]
        if not torch.isfinite(gseg).all(): print("  NONFINITE grad", name)
    model._apply()
    chk("w after adam", model.model.store.w)
    print("stats nonfinite:", (~torch.isfinite(model.optimizer.stats)).sum().item(), "step", step)
