"""Where does the bf16 path's per-attribute loss deviation come from?  (VERDICT r05 "weak" #2; not a pytest module.)

CPU study on the f64 ORACLE (test infrastructure: this script lives under tests/ because it imports ``oracle``): the model
forward of ``oracle/torch_ref.py`` restated with a rounding hook at every place where the HIP bf16 path rounds a value to
bf16 before an MFMA (csrc/block_attn.hip, heads_loss.hip, embed.hip), on the batches of
``tests/test_gpu_model.py::test_timed_shape_parity_vs_oracle`` (S = 128, D = 256, 4 blocks; c2 mix B = 4, c3 mix B = 5) and
on larger batches.  For every key: the relative deviation of its loss with ALL sites rounded (what the GPU test measures on
the device: cross-check), with ONE site rounded, and with all BUT one.

    python tests/bf16_error_budget.py [--B 4 5 32] > profiles/r06_bf16_error_budget.txt

Sites: ``weights`` (every kernel matrix of a Dense layer, bf16 shadow), ``num_in`` (the 512-wide numerical inputs of the encoder),
``y1``/``y2`` (LayerNorm outputs), ``qkv`` (q | k | v), ``p`` (unnormalised attention probabilities), ``a`` (attention output),
``h`` (FFN hidden), ``x_heads`` (the block stack's output in front of the decoder heads).  The residual stream, the LayerNorm
statistics, the softmax and every accumulator are f32 on the device and are left exact here.
"""
import argparse
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd"), os.path.join(ROOT, "tests")]

SITES = ["weights", "num_in", "y1", "qkv", "p", "a", "y2", "h", "x_heads"]


def bf16(t):
    return t.to(torch.bfloat16).to(t.dtype)


class Rounder:
    def __init__(self, on):
        self.on = set(on)

    def __call__(self, site, t):
        return bf16(t) if site in self.on else t


def forward(tr, p, ic, inputs, L, S, rnd):
    """torch_ref.model_fwd with rounding hooks (same operation order as oracle/torch_ref.py:64-157)."""
    from oracle.np_ref import LN_EPS, MASK_VALUE, NULL_VALUE, NUM_HEADS, valid_columns
    def W(name):      # "weights" rounds every kernel; "w:<group>" only the group's (encoder, attn, mlp, decoder)
        grp = "encoder" if name.startswith("encoder") else "decoder" if name.startswith("decoder") else "attn" if "/attn/" in name else "mlp"
        return rnd("w:" + name.rsplit("/", 1)[-1], rnd("w:" + grp, rnd("weights", p[name + "/kernel"])))
    dense = lambda x, name: x @ W(name) + p[name + "/bias"]
    seq_mask = tr.get_seq_mask(inputs["length"], S)
    seq = 0.0
    for key, col in valid_columns(ic).items():
        if col["type"] == "categorical":
            x = p["encoder/input_%s/embeddings" % key][inputs[key].to(torch.int64)].sum(dim=2)
        else:
            xin = inputs[key].to(torch.float64)
            is_masked, is_unused = (xin == MASK_VALUE).all(dim=2), (xin == NULL_VALUE).all(dim=2)
            special = p["encoder/input_%s_special/embeddings" % key]
            x = dense(rnd("num_in", xin), "encoder/input_%s" % key)
            x = torch.where(is_masked[..., None], special[0].expand_as(x), x)
            x = torch.where(is_unused[..., None], special[1].expand_as(x), x)
        seq = seq + x
    x = seq
    B, _, D = x.shape
    H, hd = NUM_HEADS, D // NUM_HEADS

    def ln(x, pre):
        mean = x.mean(dim=-1, keepdim=True)
        var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
        return (x - mean) / torch.sqrt(var + LN_EPS) * p[pre + "/gamma"] + p[pre + "/beta"]

    for i in range(L):
        pre = "blocks/seq2seq_%d/" % i
        y = rnd("y1", ln(x, pre + "norm1"))
        heads = lambda name: rnd("qkv", dense(y, pre + "attn/" + name)).reshape(B, S, H, hd).permute(0, 2, 1, 3)
        q, k, v = heads("dense_query"), heads("dense_key"), heads("dense_value")
        sc = (q @ k.transpose(-1, -2)) / math.sqrt(float(hd)) + -1e9 * (1.0 - seq_mask.to(x.dtype)[:, None, None, :])
        # device: P = exp2(s - max) rounded to bf16 for P V, the row sum taken from the UNROUNDED values
        e = torch.exp(sc - sc.max(dim=-1, keepdim=True).values)
        out = (rnd("p", e) @ v) / e.sum(dim=-1, keepdim=True)
        out = rnd("a", out.permute(0, 2, 1, 3).reshape(B, S, D))
        x = x + dense(out, pre + "attn/combine_heads")
        y = rnd("y2", ln(x, pre + "norm2"))
        hid = rnd("h", torch.relu(dense(y, pre + "mlp/dense_0")))
        x = x + dense(hid, pre + "mlp/dense_1")
    xh = rnd("x_heads", x)
    outputs = {}
    for key, col in valid_columns(ic).items():
        yk = dense(xh, "decoder/decoder_%s" % key)
        outputs[key] = yk.reshape(B, S, col["shape"][-1], col["input_dim"]) if col["type"] == "categorical" else yk.reshape(B, S, col["shape"][-1])
    return outputs


def key_losses(tr, ic, batch, outputs, masks, S):
    cast = {k: (v.to(torch.float64) if v.is_floating_point() else v) for k, v in batch.items()}
    total, losses, scores, _ = tr.loss_layer(ic, cast, outputs, masks, maxlen=S)
    return float(total), {k: float(v) for k, v in losses.items()}, {k: float(v) for k, v in scores.items()}


def study(mix, B, L=4, S=128, D=256):
    import test_gpu_model as tg
    ic, params, batch, modified, masks, tr, keys = tg._timed_shape_case(mix, B, S, D, L)
    p = tr.to_torch(params, torch.float64, requires_grad=False)
    with torch.no_grad():
        run = lambda on: key_losses(tr, ic, batch, forward(tr, p, ic, modified, L, S, Rounder(on)), masks, S)
        t0, l0, s0 = run([])
        # cross-check of this restatement against the oracle's own forward (must be the same numbers)
        cast = {k: (v.to(torch.float64) if v.is_floating_point() else v) for k, v in modified.items()}
        ref = key_losses(tr, ic, batch, tr.model_fwd(p, ic, cast, L, maxlen=S), masks, S)
        assert abs(ref[0] - t0) <= 1e-9 * abs(t0), (ref[0], t0)
        rel = lambda t, l: ({k: abs(l[k] - l0[k]) / max(abs(l0[k]), 1e-3 * t0) for k in l0}, abs(t - t0) / t0)
        rows = [("ALL sites (= the device's bf16 path)",) + rel(*run(SITES)[:2])]
        for s in SITES:
            rows.append(("only " + s,) + rel(*run([s])[:2]))
        for s in SITES:
            rows.append(("all but " + s,) + rel(*run([x for x in SITES if x != s])[:2]))
        for g in ("encoder", "attn", "mlp", "decoder"):
            rows.append(("only the %s kernels" % g,) + rel(*run(["w:" + g])[:2]))
        for g in ("dense_query", "dense_key", "dense_value", "combine_heads"):
            rows.append(("only the %s kernels" % g,) + rel(*run(["w:" + g])[:2]))
        acts = [x for x in SITES if x != "weights"]
        for g in ("encoder", "attn", "mlp", "decoder"):
            rows.append(("all but the %s kernels" % g,) + rel(*run(acts + ["w:" + x for x in ("encoder", "attn", "mlp", "decoder") if x != g])[:2]))
    ks = list(l0)
    print("\n== %s mix, B = %d (S = %d, D = %d, %d blocks): total data loss %.4f" % (mix, B, S, D, L, t0))
    print("   masked fields per key: " + ", ".join("%s %d" % (k, int(s0[k + "_score_den"])) for k in ks))
    print("   share of the total loss: " + ", ".join("%s %.3f" % (k, l0[k] / t0) for k in ks))
    print("%-40s %9s %9s  %s" % ("rounded sites", "total", "worst key", "  ".join("%-9s" % k[:9] for k in ks)))
    for name, kr, tot in rows:
        wk = max(kr, key=kr.get)
        print("%-40s %9.2e %9.2e  %s   <- %s" % (name, tot, kr[wk], "  ".join("%9.2e" % kr[k] for k in ks), wk))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, nargs="*", default=[4, 32])
    a = ap.parse_args()
    torch.manual_seed(0)
    print(__doc__.split("\n\n")[0])
    for B in a.B:
        study("c2", B)
    study("c3", 5)
