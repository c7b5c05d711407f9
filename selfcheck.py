"""``__graft_entry__.smoke()`` (test infrastructure next to the entry point, not part of the product
package -- it imports the oracle): one small train-step-shaped invocation of the HIP hot path on
cuda:0, checked against the CPU oracle (forward logits, per-key losses, one gradient)."""
import numpy as np
import torch


def run_smoke():
    from oracle import np_ref, torch_ref  # checker only; never on the product path
    from mfp.data.spec import make_input_columns, synthetic_batch
    from mfp.models.metrics import build_loss_keys, loss_key_names
    from mfp.models.model import Model

    ic = make_input_columns("crello")
    B, S, D, L = 4, 16, 128, 2
    params = np_ref.init_params(ic, D, L, seed=-3)
    batch = synthetic_batch(ic, B, S, seed=1, ragged=True)
    g = torch.Generator().manual_seed(0)
    masks = {k: torch.rand(B, S, generator=g) < 0.5 for k in loss_key_names(ic)}

    state = torch_ref.TrainState(params, l2=None, clipnorm=None, dtype=torch.float64)
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    info, grads = torch_ref.loss_and_grads(state, ic, b64, b64, masks, L, maxlen=S)

    model = Model(ic, num_blocks=L, latent_dim=D, dropout=0.0, l2=None, dtype="fp32", device="cuda:0")
    model.store.load_state_dict(params)
    dev = {k: v.to("cuda:0") for k, v in batch.items()}
    dmasks = {k: v.to("cuda:0") for k, v in masks.items()}
    keys = build_loss_keys(ic, model.layout.head_cols, dev, dmasks)
    loss, sums, outputs = model.forward_loss(dev, keys, training=True)
    loss.backward()
    torch.cuda.synchronize()
    want = float(info["data_loss"])
    got = float(loss)
    assert abs(got - want) < 1e-3 * max(1.0, abs(want)), (got, want)
    for k in loss_key_names(ic):
        err = (outputs[k].cpu().double() - info["outputs"][k].detach()).abs().max().item()
        assert err < 1e-3, (k, err)
    gd = model.store.grads_state_dict()
    name = "blocks/seq2seq_0/attn/dense_query/kernel"
    err = (gd[name].double() - grads[name]).abs().max().item()
    assert err < 1e-4, (name, err)
    print("smoke ok: loss %.6f (oracle %.6f), dWq max err %.2e" % (got, want, err))

    # the timed path: bf16, d_model 256 -- the activation-stationary block kernels (csrc/block_fused.hip), the
    # weight-stationary and grouped weight-gradient GEMMs, bf16 attention -- against the same oracle
    B, S, D, L = 2, 128, 256, 2
    params = np_ref.init_params(ic, D, L, seed=-5)
    batch = synthetic_batch(ic, B, S, seed=2, ragged=True)
    masks = {k: torch.rand(B, S, generator=g) < 0.3 for k in loss_key_names(ic)}
    state = torch_ref.TrainState(params, l2=None, clipnorm=None, dtype=torch.float64)
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    info, grads = torch_ref.loss_and_grads(state, ic, b64, b64, masks, L, maxlen=S)
    model = Model(ic, num_blocks=L, latent_dim=D, dropout=0.0, l2=None, dtype="bf16", device="cuda:0")
    model.store.load_state_dict(params)
    model.store.refresh_shadow()
    dev = {k: v.to("cuda:0") for k, v in batch.items()}
    dmasks = {k: v.to("cuda:0") for k, v in masks.items()}
    keys = build_loss_keys(ic, model.layout.head_cols, dev, dmasks)
    loss, sums, outputs = model.forward_loss(dev, keys, training=True)
    loss.backward()
    torch.cuda.synchronize()
    want, got = float(info["data_loss"]), float(loss)
    assert abs(got - want) < 2e-3 * max(1.0, abs(want)), (got, want)          # bf16 operands, f32 accumulation
    gd = model.store.grads_state_dict()
    for name in ("blocks/seq2seq_0/mlp/dense_0/kernel", "blocks/seq2seq_1/attn/combine_heads/kernel"):
        a, b = gd[name].double().flatten(), grads[name].flatten()
        cos = float(torch.nn.functional.cosine_similarity(a, b, dim=0))
        assert cos > 0.995, (name, cos)
    print("smoke ok (bf16, d_model 256): loss %.4f (oracle %.4f, rel %.1e)" % (got, want, abs(got - want) / abs(want)))
