"""Flat parameter storage for the MFP model.

All trainable variables of Encoder / Blocks / Decoder (reference architecture/encoder.py:72-92,
transformer.py:43-57,161-173, decoder.py:33-43) live back to back in ONE float32 buffer ``w``
with matching gradient ``g`` and Adam slots ``m``/``v`` (+ an optional bf16 shadow of ``w``
for the MFMA operands).  Reasons, all MI355X-side:

* one fused multi-tensor Adam / clipnorm / L2 pass over HBM instead of ~70 variables x ~10
  eager ops;
* one RCCL all-reduce over the flat gradient for data parallelism;
* GEMM weights are consumed in place (pointer + offset), no per-step concatenation.

Dense kernels are stored TRANSPOSED, ``[out][in]`` (Keras: ``[in][out]``), so that the fused
QKV matrix ``[3D][D]`` and the concatenated decoder heads ``[U][D]`` are contiguous row blocks
per Keras variable -- Keras' ``clipnorm`` and L2 regulariser act per variable
(architecture/utils.py:8-22; train.py:71-75).  ``state_dict()`` / ``load_state_dict()`` use
Keras names and Keras shapes (transposing on the way) -- the checkpoint surface.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from mfp.data.spec import get_valid_input_columns

WS_K = (256, 512)   # contraction lengths the weight-stationary GEMM takes with every epilogue (csrc/gemm_ws.h)
NUM_HEADS = 8  # transformer.py:147, never overridden by Blocks (transformer.py:263-270)


from mfp.models.architecture.utils import variable_l2

class Segment:
    __slots__ = ("name", "offset", "shape", "l2", "transposed", "size")

    def __init__(self, name, offset, shape, l2, transposed):
        self.name, self.offset, self.shape, self.l2, self.transposed = name, offset, tuple(shape), l2, transposed
        self.size = int(np.prod(shape))


class ModelLayout:
    """Static description of the flat buffer for one (input_columns, D, L) configuration."""

    def __init__(self, input_columns: Dict, latent_dim: int, num_blocks: int, input_dtype: str = "set",
                 context: Optional[str] = None):
        D = latent_dim
        assert D % NUM_HEADS == 0, "embedding dimension = %d should be divisible by number of heads = %d" % (
            D, NUM_HEADS)  # ValueError text of transformer.py:48-52
        assert D % 64 == 0, "latent_dim must be a multiple of 64 for the gfx950 kernels"
        self.D, self.L = D, num_blocks
        self.columns = get_valid_input_columns(input_columns, False)
        self.cat_keys = [k for k, c in self.columns.items() if c["type"] == "categorical"]
        self.num_keys = [k for k, c in self.columns.items() if c["type"] == "numerical"]
        self.segments: "OrderedDict[str, Segment]" = OrderedDict()
        self._cursor = 0

        # ---- encoder tables: categorical tables in column order, then the 2-row special tables
        self.table_start = self._cursor
        self.idx_cols: List[Tuple[str, int]] = []   # (key, feature index) per index column
        self.rowoff: List[int] = []
        rows = 0
        for k in self.cat_keys:
            c = self.columns[k]
            n_rows = c["input_dim"] + 2                       # +<MASK>, <UNUSED> (encoder.py:76)
            self._add("encoder/input_%s/embeddings" % k, (n_rows, D), True, False)
            for f in range(c["shape"][-1]):
                self.idx_cols.append((k, f))
                self.rowoff.append(rows)
            rows += n_rows
        self.special_col: Dict[str, int] = {}
        for k in self.num_keys:
            self._add("encoder/input_%s_special/embeddings" % k, (2, D), True, False)
            self.special_col[k] = len(self.idx_cols)
            self.idx_cols.append((k, -1))
            self.rowoff.append(rows)
            rows += 2
        self.table_rows = rows
        # pad the concatenated tables to a multiple of 8 rows: the table gradient is a
        # [rows][D] = onehot^T dh product on the wgrad GEMM (M % 8 == 0)
        self.table_rows_pad = (rows + 7) // 8 * 8
        if self.table_rows_pad > rows:
            self._add("encoder/_pad/embeddings", (self.table_rows_pad - rows, D), False, False)
        for k in self.num_keys:
            width = self.columns[k]["shape"][-1]
            assert width % 8 == 0
            self._add("encoder/input_%s/kernel" % k, (D, width), True, True)
            self._add("encoder/input_%s/bias" % k, (D,), True, False)
        # position token of the non-"set" input types (encoder.py:47-55: PositionEmbedding "input_const",
        # maxlen = input_dim of the length column -> maxlen + 1 rows)
        self.pos_rows = 0
        if input_dtype != "set":
            self.pos_rows = int(input_columns["length"]["input_dim"]) + 1
            self._add("encoder/input_const/embeddings", (self.pos_rows, D), True, False)
        # context token (encoder.py:96-110): one row per task (context="id") or per length (context="length");
        # the reference names both layers "input_task"
        self.context = context
        self.context_rows = 0
        if context is not None:
            if context == "id":
                from mfp.models.masking import get_task_names
                self.context_rows = len(get_task_names(input_columns))
            elif context == "length":
                self.context_rows = int(input_columns["length"]["input_dim"])
            else:
                raise NotImplementedError("context=%r: only None, 'id' and 'length' are provided (the canvas "
                                          "contexts need the canvas heads, which are off the MFP path)" % context)
            self._add("encoder/input_task/embeddings", (self.context_rows, D), True, False)

        # ---- transformer blocks
        for i in range(num_blocks):
            p = "blocks/seq2seq_%d/" % i
            for n in ("dense_query", "dense_key", "dense_value"):     # rows [3D][D] contiguous
                self._add(p + "attn/%s/kernel" % n, (D, D), True, True)
            for n in ("dense_query", "dense_key", "dense_value"):     # [3D] contiguous
                self._add(p + "attn/%s/bias" % n, (D,), True, False)
            self._add(p + "attn/combine_heads/kernel", (D, D), True, True)
            self._add(p + "attn/combine_heads/bias", (D,), True, False)
            self._add(p + "norm1/gamma", (D,), False, False)
            self._add(p + "norm1/beta", (D,), False, False)
            self._add(p + "mlp/dense_0/kernel", (2 * D, D), True, True)
            self._add(p + "mlp/dense_0/bias", (2 * D,), True, False)
            self._add(p + "mlp/dense_1/kernel", (D, 2 * D), True, True)
            self._add(p + "mlp/dense_1/bias", (D,), True, False)
            self._add(p + "norm2/gamma", (D,), False, False)
            self._add(p + "norm2/beta", (D,), False, False)

        # ---- decoder heads: rows [Upad][D] + bias [Upad]; per key a row block.  Every head starts
        # at a multiple of 8 logits columns (zero-weight pad rows in between: 16-byte bf16 / float4
        # access for the loss kernels and the GEMM epilogues); U counts the real units.
        # Categorical heads first, numerical (regression) heads behind them: the categorical logits are one
        # contiguous column range, which is what the train step evaluates on every token (the numerical heads run
        # on the compacted rows of the tokens that carry a loss: hip/functions.py DecoderLossFn).
        self.head_cols: Dict[str, Tuple[int, int]] = {}
        col, real = 0, 0
        self.heads_start = self._cursor
        pads = {}
        self.head_order = ([k for k, c in self.columns.items() if c["type"] == "categorical"]
                           + [k for k, c in self.columns.items() if c["type"] != "categorical"])
        for k in self.head_order:
            c = self.columns[k]
            units = c["shape"][-1] * c["input_dim"] if c["type"] == "categorical" else c["shape"][-1]
            self._add("decoder/decoder_%s/kernel" % k, (units, D), True, True)
            self.head_cols[k] = (col, units)
            col += units
            real += units
            pads[k] = (-col) % 8
            if pads[k]:
                self._add("decoder/_pad/%s/kernel" % k, (pads[k], D), False, False)
                col += pads[k]
        self.U = real
        self.Upad = col
        self.heads_bias_start = self._cursor
        for k in self.head_order:
            self._add("decoder/decoder_%s/bias" % k, (self.head_cols[k][1],), True, False)
            if pads[k]:
                self._add("decoder/_pad/%s/bias" % k, (pads[k],), False, False)
        self.numel = self._cursor
        assert self.numel % 8 == 0

    def _add(self, name, shape, l2, transposed):
        seg = Segment(name, self._cursor, shape, l2, transposed)
        self.segments[name] = seg
        self._cursor += seg.size

    def bucket_split(self) -> int:
        """Flat offset of the first variable of block L/2: [split, numel) = upper blocks + heads are
        complete after the first half of the backward pass, [0, split) after the second."""
        if self.L < 2:
            return 0
        return self.segments["blocks/seq2seq_%d/attn/dense_query/kernel" % (self.L // 2)].offset

    def block_offset(self, i: int) -> int:
        """Flat offset of block i's first variable (its variables, then the later blocks', then the heads' follow)."""
        return self.segments["blocks/seq2seq_%d/attn/dense_query/kernel" % i].offset

    def seg_offsets(self) -> List[int]:
        return [s.offset for s in self.segments.values()] + [self.numel]


class ParamStore:
    """Device buffers + named views for one model instance."""

    def __init__(self, layout: ModelLayout, device, compute_dtype: torch.dtype = torch.float32,
                 l2: Optional[float] = 1e-2, seed: int = 0, fp8: bool = False):
        self.layout = layout
        # fp8 mode (BASELINE config c5): e4m3 copies of the fused Q|K|V and FFN1 kernels ([out][in], as stored)
        # with one scale per copy, refreshed after every optimizer step; everything else stays bf16
        self.fp8 = bool(fp8) and compute_dtype == torch.bfloat16 and torch.device(device).type == "cuda"
        if self.fp8 and layout.D % 128 != 0:      # (one MX MFMA contracts 128 input features: csrc/gemm_fp8.hip)
            raise ValueError("dtype='fp8' needs latent_dim %% 128 == 0 (got %d)" % layout.D)
        self.shadow8, self.scale8, self._fp8_tensors = None, None, []
        self.device = torch.device(device)
        self.compute_dtype = compute_dtype
        n = layout.numel
        self.w = torch.zeros(n, dtype=torch.float32, device=device)
        self.g = torch.zeros(n, dtype=torch.float32, device=device)
        self.shadow = torch.zeros(n, dtype=torch.bfloat16, device=device) \
            if compute_dtype == torch.bfloat16 else None
        # transposed bf16 copies ([in][out]) of the Dense kernels whose input gradient runs as a
        # k-major weight-stationary product (K = out <= 512): MLP and attention output projection
        self.shadow_t = None
        self._ttable = None
        if self.shadow is not None and torch.device(device).type == "cuda":
            segs = [(s.offset, s.shape[0], s.shape[1]) for s in layout.segments.values()
                    if s.transposed and ("/mlp/" in s.name or "/combine_heads/" in s.name)
                    and (s.shape[0] in WS_K or layout.D == 512)]
            # fused Q|K|V ([3D][D], three adjacent variables) -> one [D][3D] block at the query offset (d_model 256: the
            # activation-stationary kernels of csrc/block_fused.hip; d_model 512: csrc/block_d512.hip)
            if layout.D in (256, 512):
                for i in range(layout.L):
                    q = layout.segments["blocks/seq2seq_%d/attn/dense_query/kernel" % i]
                    for j, nm in enumerate(("dense_query", "dense_key", "dense_value")):
                        sg = layout.segments["blocks/seq2seq_%d/attn/%s/kernel" % (i, nm)]
                        segs.append((sg.offset, sg.shape[0], sg.shape[1], q.offset + j * layout.D, 3 * layout.D))
            # concatenated decoder heads [Upad][D] -> [D][Upad rounded up to 128] (zero pad columns), behind the end of
            # the buffer: the activation-stationary input gradient of the heads (csrc/block_fused.hip)
            self._heads_t = None
            if layout.D in (256, 512):      # (d_model 512: csrc/block_d512.hip, mfp_dense_n512_lda)
                ldw = (layout.Upad + 127) // 128 * 128
                segs.append((layout.heads_start, layout.Upad, layout.D, n, ldw))
                self._heads_t = (n, ldw)
            if segs:
                from mfp.hip import ops
                self.shadow_t = torch.zeros(n + (layout.D * self._heads_t[1] if self._heads_t else 0), dtype=torch.bfloat16,
                                            device=device)
                self._ttable = ops.TransposeTable(segs, device)
        if self.fp8:
            self.shadow8 = torch.zeros(n, dtype=torch.uint8, device=device)
            D = layout.D
            for i in range(layout.L):
                q = layout.segments["blocks/seq2seq_%d/attn/dense_query/kernel" % i]
                f = layout.segments["blocks/seq2seq_%d/mlp/dense_0/kernel" % i]
                self._fp8_tensors += [(q.offset, 3 * D * D), (f.offset, f.size)]
            self.scale8 = torch.zeros(n // 32 + 1, dtype=torch.uint8, device=device)      # e8m0 block scales, element offset / 32
        self.l2 = l2
        # (s.l2 is False for the zero pad rows / special rows that are not Keras variables)
        self.seg_l2 = torch.tensor([variable_l2(s.name, l2) if s.l2 else 0.0 for s in layout.segments.values()],
                                   dtype=torch.float32, device=device)
        self.rowoff = torch.tensor(layout.rowoff, dtype=torch.int32, device=device)
        # autograd anchor: the custom Functions need one differentiable input even when the data
        # inputs are integer indices; parameter gradients bypass autograd and land in ``g``.
        self.anchor = torch.zeros((), dtype=torch.float32, device=device, requires_grad=True)
        self._scratch = {}
        self.init_keras(seed)

    def scratch(self, name: str, shape, dtype) -> torch.Tensor:
        """Persistent zero-initialised device buffer (allocated once per shape), e.g. the dlogits
        matrix whose padding columns must stay zero without a 90 MB fill every step."""
        key = (name, tuple(shape), dtype)
        buf = self._scratch.get(key)
        if buf is None:
            buf = torch.zeros(tuple(shape), dtype=dtype, device=self.device)
            self._scratch[key] = buf
        return buf

    # ------------------------------------------------------------------ views
    def _view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        s = self.layout.segments[name]
        return buf[s.offset:s.offset + s.size].view(s.shape)

    def weight(self, name: str) -> torch.Tensor:
        return self._view(self.w, name)

    def grad(self, name: str) -> torch.Tensor:
        return self._view(self.g, name)

    def cw(self, name: str, rows: Optional[int] = None) -> torch.Tensor:
        """Compute-dtype view of a GEMM weight starting at ``name`` (optionally spanning
        ``rows`` rows across the following contiguous variables, e.g. the fused QKV)."""
        s = self.layout.segments[name]
        buf = self.shadow if self.shadow is not None else self.w
        if rows is None:
            return buf[s.offset:s.offset + s.size].view(s.shape)
        cols = s.shape[-1]
        return buf[s.offset:s.offset + rows * cols].view(rows, cols)

    def span(self, buf: torch.Tensor, name: str, count: int, cols: Optional[int] = None):
        s = self.layout.segments[name]
        t = buf[s.offset:s.offset + count]
        return t.view(-1, cols) if cols else t

    def tables(self, buf=None) -> torch.Tensor:
        buf = self.w if buf is None else buf
        L = self.layout
        return buf[L.table_start:L.table_start + L.table_rows * L.D].view(L.table_rows, L.D)

    def cwt(self, name: str) -> Optional[torch.Tensor]:
        """[in][out] bf16 view of the Dense kernel ``name`` (None when no transposed shadow is kept)."""
        if self.shadow_t is None:
            return None
        s = self.layout.segments[name]
        if name.endswith("attn/dense_query/kernel"):     # the fused Q|K|V block, [D][3D]
            if self.layout.D not in (256, 512):
                return None
            return self.shadow_t[s.offset:s.offset + 3 * s.size].view(s.shape[1], 3 * s.shape[0])
        if not (s.transposed and ("/mlp/" in name or "/combine_heads/" in name) and (s.shape[0] in WS_K or self.layout.D == 512)):
            return None
        return self.shadow_t[s.offset:s.offset + s.size].view(s.shape[1], s.shape[0])

    def heads_t(self) -> Optional[torch.Tensor]:
        """[D][ldw] bf16: the concatenated head kernels transposed, zero beyond column Upad (None when not kept)."""
        if self.shadow_t is None or getattr(self, "_heads_t", None) is None:
            return None
        off, ldw = self._heads_t
        return self.shadow_t[off:off + self.layout.D * ldw].view(self.layout.D, ldw)

    def tables_padded(self, buf=None) -> torch.Tensor:
        """[table_rows_pad][D] view: the tables plus the zero pad rows behind them."""
        buf = self.w if buf is None else buf
        L = self.layout
        return buf[L.table_start:L.table_start + L.table_rows_pad * L.D].view(L.table_rows_pad, L.D)

    def refresh_transposed(self):
        if self.shadow_t is not None:
            from mfp.hip import ops
            ops.transpose_cast_bf16(self.w, self.shadow_t, self._ttable)
        self.refresh_fp8()

    def refresh_fp8(self):
        """Re-quantise the MX fp8 weight copies (e4m3 elements + one e8m0 scale per 32 input features) from the f32
        master weights.  Kernels are stored [out][in], so a block is 32 consecutive elements of the flat buffer; segment
        offsets are multiples of 32 (ModelLayout pads), hence the scale of element e sits at e // 32."""
        if not self.fp8:
            return
        from mfp.hip import ops
        K = self.layout.D
        for off, n in self._fp8_tensors:
            assert off % 32 == 0 and K % 32 == 0
            ops.quantize_mxfp8(self.w[off:off + n], n // K, K, self.shadow8[off:off + n], self.scale8[off // 32:(off + n) // 32])

    def w8(self, name: str, rows: int):
        """(fp8 view [rows][in], e8m0 scales [rows][in / 32]) of the kernel starting at ``name`` (fp8 mode only)."""
        s = self.layout.segments[name]
        cols = s.shape[-1]
        return (self.shadow8[s.offset:s.offset + rows * cols].view(rows, cols),
                self.scale8[s.offset // 32:s.offset // 32 + rows * cols // 32].view(rows, cols // 32))

    def refresh_shadow(self):
        if self.shadow is not None:
            from mfp.hip import ops
            ops.cast_bf16(self.w, self.shadow)
            self.refresh_transposed()

    # ------------------------------------------------------------------ init / (de)serialise
    def init_keras(self, seed: int = 0):
        """[TF-EXT] Keras defaults: Dense glorot_uniform + zero bias; Embedding U(-0.05, 0.05);
        LayerNormalization gamma 1 / beta 0."""
        rng = np.random.default_rng(seed)
        host = np.zeros(self.layout.numel, dtype=np.float32)
        for s in self.layout.segments.values():
            if "/_pad/" in s.name:
                continue
            if s.name.endswith("/embeddings"):
                v = rng.uniform(-0.05, 0.05, size=s.shape)
            elif s.name.endswith("/kernel"):
                limit = np.sqrt(6.0 / (s.shape[0] + s.shape[1]))
                v = rng.uniform(-limit, limit, size=s.shape)
            elif s.name.endswith("/gamma"):
                v = np.ones(s.shape)
            else:
                v = np.zeros(s.shape)
            host[s.offset:s.offset + s.size] = v.reshape(-1)
        self.w.copy_(torch.from_numpy(host))
        if self.w.is_cuda:
            self.refresh_shadow()

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        """Keras-named, Keras-shaped (kernels ``[in][out]``) CPU tensors."""
        out = OrderedDict()
        w = self.w.detach().cpu()
        for s in self.layout.segments.values():
            if "/_pad/" in s.name:
                continue
            t = w[s.offset:s.offset + s.size].view(s.shape)
            out[s.name] = t.t().contiguous() if s.transposed else t.clone()
        return out

    def load_state_dict(self, state: Dict[str, "torch.Tensor | np.ndarray"], strict: bool = True):
        host = self.w.detach().cpu().clone()
        missing = []
        for s in self.layout.segments.values():
            if "/_pad/" in s.name:
                continue
            if s.name not in state:
                missing.append(s.name)
                continue
            t = torch.as_tensor(np.asarray(state[s.name]), dtype=torch.float32)
            if s.transposed:
                t = t.t()
            if tuple(t.shape) != s.shape:
                raise ValueError("shape mismatch for %s: %s vs %s" % (s.name, tuple(t.shape), s.shape))
            host[s.offset:s.offset + s.size] = t.reshape(-1)
        if strict and missing:
            raise KeyError("missing variables: %s" % missing)
        self.w.copy_(host)
        if self.w.is_cuda:
            self.refresh_shadow()

    def grads_state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        out = OrderedDict()
        g = self.g.detach().cpu()
        for s in self.layout.segments.values():
            if "/_pad/" in s.name:
                continue
            t = g[s.offset:s.offset + s.size].view(s.shape)
            out[s.name] = t.t().contiguous() if s.transposed else t.clone()
        return out
