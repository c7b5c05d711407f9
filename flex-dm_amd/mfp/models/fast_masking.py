"""Train-step fast path for ``preprocess_for_train`` (reference models/mfp.py:95-138): one fused
HIP launch (``mfp_mask_tokens``) instead of ~300 element-wise ops.  Same semantics and
probabilities as ``mfp.models.masking`` (which stays the reference-shaped API used by eval and by
the tests); a different -- counter-based -- random stream.
"""
from typing import Dict, List

import torch

from mfp import dp
from mfp.data.spec import get_attribute_groups
from mfp.hip import ops

MASK_SEED_SALT = 0x9E3779B97F4A7C15


class FusedMasker:
    def __init__(self, input_columns: Dict, layout, store, seed: int):
        self.layout, self.store = layout, store
        self._seed = int(seed)
        groups = get_attribute_groups(input_columns.keys())
        group_of = {k: gi for gi, keys in enumerate(groups.values()) for k in keys}
        self.cols: List[dict] = []
        col_pos = {}
        for pos, (k, f) in enumerate(layout.idx_cols):
            col_pos.setdefault(k, pos)
        for k, c in layout.columns.items():
            d = dict(key=k, is_numerical=c["type"] != "categorical", group=group_of.get(k, -100),
                     idx_col=col_pos[k])
            if c["type"] == "categorical":
                d.update(n_feat=c["shape"][-1], input_dim=c["input_dim"])
            else:
                d.update(n_feat=c["shape"][-1], input_dim=0)
            if "loss_condition" in c:
                cond = c["loss_condition"]
                d.update(cond_key=cond["key"], cond_bits=sum(1 << i for i, f in enumerate(cond["mask"]) if f))
            self.cols.append(d)

    @property
    def seed(self) -> int:
        """Evaluated per call: the process group may be initialised after the model is built."""
        return (dp.rank_seed(self._seed) ^ MASK_SEED_SALT) & 0xFFFFFFFFFFFFFFFF

    def __call__(self, batch: Dict[str, torch.Tensor], tasks: torch.Tensor, nvalid: torch.Tensor, B: int, S: int,
                 step_ptr):
        L, dev, cdt = self.layout, tasks.device, self.store.compute_dtype
        T = B * S
        idx_all = torch.empty((T, len(L.idx_cols)), dtype=torch.int32, device=dev)
        masks_u8 = torch.empty((len(self.cols), T), dtype=torch.uint8, device=dev)
        descr, codes, xs, keep = [], [], [], []
        for i, c in enumerate(self.cols):
            src = batch[c["key"]]
            src = (src.to(torch.float32) if c["is_numerical"] else src.to(torch.int32)).contiguous()
            keep.append(src)
            d = dict(is_numerical=c["is_numerical"], n_feat=c["n_feat"], input_dim=c["input_dim"],
                     group=c["group"], src=src, idx_col=c["idx_col"], mask_out=masks_u8[i])
            if "cond_key" in c:
                ck = batch[c["cond_key"]].to(torch.int32).contiguous()
                keep.append(ck)
                d.update(cond_idx=ck, cond_stride=ck.shape[-1], cond_bits=c["cond_bits"])
            if c["is_numerical"]:
                x = torch.empty((T, c["n_feat"]), dtype=cdt, device=dev)
                code = torch.empty((T,), dtype=torch.uint8, device=dev)
                d.update(x_out=x, rowcode=code)
                xs.append(x)
                codes.append(code)
            descr.append(d)
        ops.mask_tokens(descr, idx_all, nvalid, tasks if tasks.dtype == torch.int32 else tasks.to(torch.int32), B, S,
                        self.seed, 0, step_ptr, cdt)
        masks = {c["key"]: masks_u8[i].view(B, S) for i, c in enumerate(self.cols)}
        return idx_all, codes, xs, masks
