#!/usr/bin/env python
"""Generates tests/golden/*.npz from the numpy-f64 oracle (NOT from the reference: TensorFlow is not
installable here and the reference holds no golden vectors; see oracle/np_ref.py "PARITY UNPINNED").

Each fixture: weights, a ragged batch, the masked inputs (every token type <MASK>/<UNUSED>/random
present), MFP masks, expected logits, per-key losses, the L2 term, all gradients and the
parameters after one clipnorm + Keras-Adam step.

    python tests/golden/make_fixtures.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "flex-dm_amd"))

from oracle import np_ref, torch_ref  # noqa: E402
from mfp.data.spec import make_input_columns, synthetic_batch  # noqa: E402
from mfp.models import masking  # noqa: E402
from mfp.models.architecture.mask import get_seq_mask  # noqa: E402

CONFIGS = [dict(name="crello_d8_l2", dataset="crello", B=3, S=5, D=8, L=2, seed=11),
           dict(name="rico_d16_l1", dataset="rico", B=4, S=6, D=16, L=1, seed=12),
           # round 6: a size the HIP path runs (d_model 128 is its smallest: 8 heads of 16), so that a `-m gpu` test reads a
           # committed fixture too (tests/test_gpu_model.py::test_hip_f32_path_vs_committed_golden_fixture).  `summary`: the
           # 230 k parameters are regenerated from the seed (np_ref.init_params) and pinned by per-variable sums; gradients and
           # the Adam step are pinned by per-variable norms and projections on seeded +-1 vectors -- 60 KB instead of 3 MB
           dict(name="rico_d128_l1_summary", dataset="rico", B=3, S=8, D=128, L=1, seed=13, summary=True)]


def sign_vector(name, n):
    """The +-1 vector a variable's gradient / Adam delta is projected on (seeded by the variable's name)."""
    import zlib
    return np.random.default_rng(zlib.crc32(name.encode())).integers(0, 2, n).astype(np.float64) * 2.0 - 1.0


def main():
    for cfg in CONFIGS:
        ic = make_input_columns(cfg["dataset"])
        nd = {k: v for k, v in ic.items() if not v.get("demo_only")}
        B, S, D, L = cfg["B"], cfg["S"], cfg["D"], cfg["L"]
        params = np_ref.init_params(ic, D, L, seed=-cfg["seed"])
        batch = synthetic_batch(ic, B, S, seed=cfg["seed"], ragged=True)
        gen = torch.Generator().manual_seed(cfg["seed"])
        seq_mask = get_seq_mask(batch["length"], maxlen=S)
        filtered = masking.filter_padding(batch, nd, seq_mask)
        modified, masks = {}, {}
        for k, c in nd.items():
            if not c["is_sequence"]:
                modified[k] = filtered[k]
                continue
            m = seq_mask & (torch.rand(B, S, generator=gen) < 0.5)
            r = torch.rand(B, S, generator=gen)
            x = masking.apply_token(filtered[k], c, m & (r < 0.7), "masked")
            x = masking.apply_token(x, c, m & (r >= 0.7) & (r < 0.85), "random", gen)
            modified[k], masks[k] = x, m
        modified["length"] = batch["length"]
        meta = dict(dataset=cfg["dataset"], B=B, S=S, D=D, L=L, l2=1e-2, lr=1e-2)
        nb = {k: v.numpy() for k, v in batch.items()}
        nm = {k: v.numpy() for k, v in modified.items()}
        nk = {k: v.numpy() for k, v in masks.items()}
        out = np_ref.model_fwd(params, ic, nm, L, maxlen=S)
        lt, losses, scores, metrics = np_ref.loss_layer(ic, nb, out, nk, S)
        state = torch_ref.TrainState(params, lr=meta["lr"], l2=meta["l2"], clipnorm=1.0, dtype=torch.float64)
        info, grads = torch_ref.loss_and_grads(state, ic, batch, modified, masks, L, maxlen=S)
        assert abs(float(info["data_loss"]) - lt) < 1e-9 * abs(lt)
        torch_ref.apply_gradients(state, grads)
        arrays = {"meta": np.array(json.dumps(meta)), "reg_loss": np.array(np_ref.l2_loss(params, meta["l2"]))}
        if cfg.get("summary"):
            meta["param_seed"] = -cfg["seed"]
            arrays["meta"] = np.array(json.dumps(meta))
            for k, v in params.items():
                g = grads[k].numpy().astype(np.float64).reshape(-1)
                d = (state.p[k].detach().numpy().astype(np.float64) - v.astype(np.float64)).reshape(-1)
                sv = sign_vector(k, g.size)
                # [sum of the parameters, |g|, g . s, |adam delta|, delta . s]
                arrays["summary:" + k] = np.array([v.astype(np.float64).sum(), np.linalg.norm(g), g @ sv, np.linalg.norm(d), d @ sv])
        for k, v in ([] if cfg.get("summary") else params.items()):
            arrays["param:" + k] = v
            arrays["grad:" + k] = grads[k].numpy().astype(np.float32)          # f32 storage keeps the
            arrays["adam1:" + k] = state.p[k].detach().numpy().astype(np.float32)  # fixtures small
        for k, v in nb.items():
            arrays["batch:" + k] = v
        for k, v in nm.items():
            arrays["modified:" + k] = v.astype(np.float32) if v.dtype.kind == "f" else v
        for k, v in nk.items():
            arrays["mask:" + k] = v
        for k, v in out.items():
            arrays["logits:" + k] = v
        for k, v in losses.items():
            arrays["loss:" + k] = np.array(v)
        path = os.path.join(HERE, cfg["name"] + ".npz")
        np.savez_compressed(path, **arrays)
        print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
