"""``torch.autograd.Function`` wrappers: one per reference Layer on the hot path.

Autograd only chains the activation gradient (element stream ``x``) between these nodes;
parameter gradients are written by the kernels straight into ``ParamStore.g`` (each variable is
produced exactly once per step, so the writes are assignments, never accumulations).

    EncoderFn      <-> Encoder.call          (reference architecture/encoder.py:147-199)
    BlockFn        <-> DeepSVGBlock.call     (architecture/transformer.py:211-229)
    DecoderFn      <-> Decoder.call          (architecture/decoder.py:95-111)
    DecoderLossFn  <-> Decoder.call + LossLayer.call fused for the train step
                                             (decoder.py:95-111 + models/metrics.py:213-299)
"""
from __future__ import annotations

from typing import Dict, List, Optional

import os

import torch

from . import ops

NUM_HEADS = 8
# bf16 path: the weight gradients of a block (of the heads, of the encoder) as ONE grouped launch with the
# split-K reduction inside it (csrc/gemm_wgg.h); 0 = one mfp_gemm + reduce kernel per product (A/B switch)
WGRAD_GROUP = os.environ.get("MFP_WGRAD_GROUP", "1") == "1"
# ... with the split-K reduction of every grouped launch of a backward pass (or of a data-parallel bucket) deferred into ONE
# launch at its end (mfp_wgrad_group_partial + mfp_wgrad_reduce: ~10 us less per grouped launch, bit-identical
# gradients); 0 = every grouped launch reduces in place (tickets + last arriver)
WGRAD_DEFER = os.environ.get("MFP_WGRAD_DEFER", "1") == "1"
# bf16 path, d_model 256: LN2 + FFN1 + ReLU + FFN2 + dropout + residual of a block as ONE launch, and the two
# input-gradient products of the same half as one launch (csrc/block_fused.hip); 0 = ln_fwd + two products,
# two dgrad products (A/B switch)
MLP_FUSE = os.environ.get("MFP_MLP_FUSE", "1") == "1"
# bf16 path, d_model 256, documents of exactly 128 positions: LN1 + Q|K|V + attention + output projection + dropout +
# residual of a block as ONE launch (csrc/block_attn.hip: a 128-row tile is a document); 0 = the three launches
ATTN_BLOCK = os.environ.get("MFP_ATTN_BLOCK", "1") == "1"
# ... and the MLP half behind it on the same tile: the whole block forward in one launch; 0 = attention half + mlp_fused
BLOCK_FWD = os.environ.get("MFP_BLOCK_FWD", "1") == "1"
# ... and its inference form (nothing saved for a backward pass) for the callers that never differentiate; 0 = the training form
BLOCK_INFER = os.environ.get("MFP_BLOCK_INFER", "1") == "1"
# bf16 residual-gradient stream, d_model 256: the backward of LN2 in the epilogue of the MLP half's input-gradient launch
# (mfp_mlp_bwd_ln: dy2 never leaves the CU); 0 = mlp_fused_bwd + layernorm_bwd (A/B switch)
MLP_BWD_LN = os.environ.get("MFP_MLP_BWD_LN", "1") == "1"
# ... and the backward of LN1 in the epilogue of the attention half's (mfp_attn_block_bwd_ln: dy1 never leaves the CU)
ATTN_BWD_LN = os.environ.get("MFP_ATTN_BWD_LN", "1") == "1"
# ... and the one-launch block forward stashes x-hat = (x - mean) rstd (bf16) in the place of LN(x): the LayerNorm-backward
# epilogues read it instead of the f32 rows (-134 MB per c2 step), the Q|K|V / FFN1 weight gradients are formed from it and
# corrected by gamma / beta in the split-K reduction (mfp_block_fwd_xhat, mfp_wgrad_job::n_affine); 0 = LN(x) stash (A/B switch)
XHAT_STASH = os.environ.get("MFP_XHAT_STASH", "1") == "1"


def _xhat_ok(ctx) -> bool:
    """Forward-time decision for the x-hat stash: the train step whose backward pass runs the LayerNorm backward in the
    input-gradient launches and leaves its weight gradients to the deferred grouped reduction.  (A backward pass that finds
    one of these missing still works: the stand-alone LayerNorm backward reads x, and y is rebuilt from x-hat for a weight
    gradient outside the deferred reduction.)"""
    return (XHAT_STASH and MLP_BWD_LN and ATTN_BWD_LN and WGRAD_GROUP and ctx.wgrad_pending is not None and _res16_ok(ctx)
            and ctx.store.layout.D in (256, 512))
# the gradient of the residual stream (what one block's backward hands to the next) in bf16 instead of f32 on the bf16
# train step: every LayerNorm backward then reads and writes 0.5 KB instead of 1 KB per element for it.  autograd sees
# stride-0 placeholders of the activations' dtype and shape; the real gradient travels in StepCtx.res_grad.  ON by
# default since round 4 (-25 us per c2 step; gradient cosine against the f64 oracle 0.99972 vs 0.99975 with the f32
# stream, tests/test_gpu_model.py), at d_model 512 since round 5 (the heads' input-gradient product writes bf16,
# mfp_dropout_bwd_res16 masks it for the last block); "0" = f32 stream
RES_GRAD_BF16 = os.environ.get("MFP_RES_GRAD_BF16", "1") == "1"


def _res16_ok(ctx) -> bool:
    """The bf16 train step whose only consumers of the residual gradient are the LayerNorm backward kernels and the
    encoder's grouped weight-gradient launch (which reads its bf16 copy anyway)."""
    L = ctx.store.layout
    return (RES_GRAD_BF16 and ctx.cdt == torch.bfloat16 and ctx.training and ctx.tail["fuse"] and WGRAD_GROUP and MLP_FUSE
            and L.table_rows_pad <= 1024 and L.D in (256, 512) and L.L > 0)


# heads forward + LossLayer + heads input gradient in one launch (csrc/heads_loss.hip); "0" = four launches
HEADS_FUSED = os.environ.get("MFP_HEADS_FUSED", "1") == "1"
# the attention half's input gradients (da, attention backward, dy1) in one launch (csrc/block_attn_bwd.hip); "0" = three
# unset: when the documents fill the chip (one workgroup = one document per CU); with fewer documents than CUs (c4: 128 per
# GPU) the three launches, which split a document over more workgroups, are faster (1.187 vs 1.210 ms per step)
ATTN_BLOCK_BWD = os.environ.get("MFP_ATTN_BLOCK_BWD", "")


# the grouped weight-gradient launches of consecutive blocks as ONE launch (the step with the deferred reduction)
# MFP_WGRAD_PAIR = blocks per launch: 4 (default: 16 products, 128 tiles = 64 macro tiles x 4 k-slices fill the chip with a quarter
# of the split-K slab bytes of one launch per block), 2 (the first form of round 6), 1 / 0 = one launch per block
WGRAD_PAIR = int(os.environ.get("MFP_WGRAD_PAIR", "4") or 0)


def _wgrad_pair_on(ctx, D) -> bool:
    # (the data-parallel step flushes what is held at the end of every backward segment -- StepCtx.flush_ln_jobs, in front of the
    #  bucket's all-reduce -- so a launch never straddles a bucket boundary: one launch per block under MFP_DP_BUCKETS=blocks,
    #  the blocks of a half under "halves")
    return WGRAD_PAIR > 1 and D in (256, 512) and ctx.wgrad_pending is not None


def _attn_block_bwd_on(ctx) -> bool:
    if ATTN_BLOCK_BWD != "":
        return ATTN_BLOCK_BWD == "1"
    return ctx.T // 128 >= ops.cu_count(ctx.store.w.device)      # (128-row tiles: documents at S = 128, document pairs at S = 64)
FUSE_MAX_T = 1 << 20     # the activation-stationary kernels address rows with 32-bit byte offsets
# the one-launch block forward on HALF-document tiles (two four-wave workgroups per document, mfp_block_fwd_xhat_half); unset:
# when two workgroups per document still fit the chip in one round (c4: 128 documents per GPU on 256 CUs); "0" / "1" = A/B
BLOCK_HALF = os.environ.get("MFP_BLOCK_HALF", "")


# the three-launch attention route of batches with fewer 128-row tiles than CUs: LN1 backward in the epilogue of dy1 = dqkv Wqkv
# on 64-row tiles (mfp_dgrad_qkv_ln_half); 0 = mfp_dgrad_qkv + mfp_layernorm_bwd_xhat (A/B switch)
DGRAD_LN_HALF = os.environ.get("MFP_DGRAD_LN_HALF", "1") == "1"
# ... and the MLP half's input-gradient launch (+ LN2 backward) on half tiles (mfp_mlp_bwd_ln_half); unset: as above
MLP_BWD_HALF = os.environ.get("MFP_MLP_BWD_HALF", "")


def _mlp_bwd_half_on(ctx, T: int) -> bool:
    if MLP_BWD_HALF != "":
        return MLP_BWD_HALF == "1"
    return 2 * (T // 128) <= ops.cu_count(ctx.store.w.device)


def _block_half_on(ctx, B: int, S: int) -> bool:
    """Half tiles: half a document at S = 128, ONE document at S = 64 (two documents per 128-row tile: the datasets' shape --
    at the reference's default batch of 256 documents that is 128 tiles on 256 CUs)."""
    if S not in (64, 128):
        return False
    if BLOCK_HALF != "":
        return BLOCK_HALF == "1"
    return 2 * (B * S // 128) <= ops.cu_count(ctx.store.w.device)


def _doc_tile_ok(B: int, S: int, T: int) -> bool:
    """Shapes the document-tile kernels take (csrc/block_attn.hip, block_attn_bwd.hip): a 128-row tile is one document of 128
    positions or two documents of 64 (the datasets' sequences are at most 51 positions long: --seq_len 64)."""
    return T == B * S and (S == 128 or (S == 64 and B % 2 == 0))


def _fused_ok(ctx, D) -> bool:
    """One predicate for every activation-stationary kernel of csrc/block_fused.hip (bf16, d_model 256,
    token count inside the kernels' 32-bit row offsets); otherwise the tiled / weight-stationary path runs."""
    return MLP_FUSE and ctx.cdt == torch.bfloat16 and D == 256 and ctx.T <= FUSE_MAX_T


# bf16 path, d_model 512 (BASELINE config c5): LN1 + Q|K|V and LN2 + FFN1 as one launch each, the products with a d_model-wide
# output (attention output projection, FFN2, the three input gradients) on the row-owning kernel (csrc/block_d512.hip);
# 0 = ln_fwd + the generic weight-stationary / LDS-tiled products
D512_FUSE = os.environ.get("MFP_D512_FUSE", "1") == "1"
# ... and the LayerNorm backward in the epilogue of the products that form a LayerNorm output's gradient (dy2 = dh W1, dy1 = dqkv
# Wqkv: mfp_dense_n512_lnb -- dy never reaches HBM, no stand-alone ln_bwd launch; x-hat stash + bf16 residual-gradient stream);
# 0 = mfp_dense_n512 + mfp_layernorm_bwd_xhat (A/B switch)
D512_LN_BWD = os.environ.get("MFP_D512_LN_BWD", "1") == "1"


# fp8 mode, d_model 512, MEASUREMENT switches (tests/test_gpu_model.py::test_c5_fp8_deviation_and_training):
# which forward products run as MX fp8 products ("qkv,ffn1" = the mode; "ffn1": Q|K|V, whose result feeds a softmax, stays
# bf16), and "weights only" = bf16 products on the MX-quantised weights (e4m3 weights x bf16 activations)
FP8_PRODUCTS = set(os.environ.get("MFP_FP8_PRODUCTS", "qkv,ffn1").split(","))
FP8_WEIGHTS_ONLY = os.environ.get("MFP_FP8_WEIGHTS_ONLY", "0") == "1"


def _w8_as_bf16(st, name: str, rows: int) -> torch.Tensor:
    """The MX-quantised copy of a kernel, dequantised to bf16 [rows][in] (exact: e4m3 x 2^e fits bf16)."""
    q, sc = st.w8(name, rows)
    v = q.view(torch.float8_e4m3fn).to(torch.float32).view(rows, -1, 32)
    return (v * torch.exp2(sc.to(torch.float32) - 127.0)[..., None]).view(rows, -1).to(torch.bfloat16).contiguous()


def _fused512_ok(ctx, D, mx_forward: bool = False) -> bool:
    """d_model 512 kernels of csrc/block_d512.hip.  ``mx_forward``: the LN + Dense forward launches, which the fp8 mode replaces
    by ln_fwd + its MX block-scaled product; the other products (output projection, FFN2, the input gradients) are bf16 in
    both modes."""
    return (D512_FUSE and ctx.cdt == torch.bfloat16 and D == 512 and ctx.T <= (1 << 19)
            and not (mx_forward and ctx.store.fp8))


def _ln_dense(ctx, x, gamma, beta, W, T, N, D, bias, relu=False, w8=None):
    """``Dense(LayerNormalization(x))`` of a DeepSVG block (transformer.py:216-217 / 222-223): returns
    (out, y = LN(x) in the compute dtype, mean, rstd).  ``w8`` = (fp8 kernel, its e8m0 block scales): fp8 mode."""
    cdt = ctx.cdt
    if w8 is not None:      # MX block-scaled e4m3 operands (csrc/gemm_fp8.hip); y stays bf16 for the backward pass
        y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, cdt)
        return ops.gemm_mxfp8(y, w8[0], w8[1], T, N, D, bias=bias, relu=relu), y, mean, rstd
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, cdt)
    out = ops.gemm(y, W, T, N, D, a_kmajor=True, b_kmajor=True, bias=bias, relu=relu, out_dtype=cdt)
    return out, y, mean, rstd


class StepCtx:
    """Per-call constants shared by the Functions of one forward pass."""

    def __init__(self, store, B: int, S: int, nvalid: torch.Tensor, training: bool, dropout: float,
                 seed: int, step_ptr: Optional[torch.Tensor], side_stream=None):
        self.store, self.B, self.S, self.T = store, B, S, B * S
        # side_stream: one HIP stream or a list; [0] carries the per-block weight gradients, the others
        # let the independent gradient products at the very end of the backward pass run side by side
        self.sides = list(side_stream) if isinstance(side_stream, (list, tuple)) else ([side_stream] if side_stream is not None else [])
        self.side = self.sides[0] if self.sides else None
        self.dh_c = None    # compute-dtype copy of the encoder output gradient (block 0's LN1 backward)
        self.onehot = None  # one-hot count matrix of the index columns (built during the forward pass)
        self.mid = None     # activation entering block L/2 (set by Blocks; see MFP.capture_train_step)
        self.cuts = {}      # block index -> the activation entering that block (every block > 0; the data-parallel
                            # step cuts its backward pass at some of them, MFP.capture_train_step)
        # pending LayerNorm parameter-gradient reductions (flush_ln_jobs); None = reduce in line
        self.ln_jobs = []
        # grouped weight-gradient launches whose split-K reduction is still pending (flush_ln_jobs); None = reduce in place
        # (side streams: the groups' slab buffers are per stream, and a deferred reduction would have to join them first)
        self.wgrad_pending = [] if (WGRAD_DEFER and not self.sides) else None
        self.loss_sort = None   # RICO position-sorted loss: dict(flag, labels, heads, ignore_sort)
        self.handoff = {}   # block index -> pre-masked bf16 gradient of its second Dropout (fused LN bwd)
        self.res_grad = None   # the bf16 gradient of the residual stream on its way down (RES_GRAD_BF16), else None
        # train-step tail fusions (shared with the context-token view of this step): "fuse" (set by
        # Model.forward_loss when the last block feeds the heads directly), "x_c" = (x2, its bf16 copy written by
        # the last block's MLP kernel), "bias_wgg" = blocks whose dense_1 bias gradient comes from the grouped
        # weight-gradient launch (their masked gradient was produced without column sums), "sums" = the flat
        # [3 nkeys + 1] loss accumulator zeroed by the step prologue
        self.tail = {"fuse": False, "x_c": None, "bias_wgg": set(), "sums": None}
        self.nvalid = nvalid
        self.training = training
        self.p = float(dropout) if training else 0.0
        self.seed = int(seed)
        self.step_ptr = step_ptr
        self.cdt = store.compute_dtype

    def with_context_token(self) -> "StepCtx":
        """The same step seen by the blocks when a context token is prepended (encoder.py:245-248): one more
        position per document, one more valid key.  Shares the pending-reduction list, the hand-off table
        and the streams with the encoder / decoder context."""
        import copy
        c = copy.copy(self)
        c.S, c.T = self.S + 1, self.B * (self.S + 1)
        c.nvalid = (self.nvalid + 1).contiguous()
        return c

    # MFP_SIDE_STREAMS=2: only a block's grouped weight-gradient launch leaves the main stream, and only for as long as
    # that block's LN1 backward (an HBM-bound streaming kernel whose workgroups fit beside the weight-gradient ones) runs:
    # BlockFn joins before it returns.  Every other on_side call joins at once (= in line).
    OVERLAP = False      # (MFP_SIDE_STREAMS was measured slower in rounds 1-4 in every form and is gone: one stream)

    def on_side(self, fn, *tensors, which: int = 0, hold: bool = False):
        """Run ``fn`` (weight-gradient GEMMs: off the critical path, only Adam needs them) on the
        side HIP stream, forked after everything enqueued so far on the current stream.  Every
        kernel of the step leaves most of a CU idle (profiles/r01_gemm_qkv_timeline.txt), so the
        wgrads overlap with the dgrad / attention / LayerNorm chain.  ``join_side`` must be called
        before the gradients are consumed."""
        if self.side is None:
            return fn()
        side = self.sides[which % len(self.sides)]
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fn()
        for t in tensors:   # their memory must not be recycled by main-stream allocations too early
            t.record_stream(side)
        if self.OVERLAP and not hold:
            main.wait_stream(side)

    def join_held(self):
        if self.OVERLAP and self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def flush_ln_jobs(self):
        """Sum the LayerNorm gamma / beta (/ fused bias) gradient partials of every layer handled so
        far in one launch.  Called where the gradients are needed: at the end of the backward pass and,
        in the data-parallel split step, before the upper bucket is all-reduced."""
        if self.ln_jobs:
            ops.reduce_partials_batch(self.ln_jobs)
            self.ln_jobs.clear()      # (in place: the context-token view of the step shares the list)
        held = self.tail.get("wgrad_held")
        if held is not None:      # (a block whose weight-gradient launch waited for a partner that did not come: a cut backward pass)
            self.tail["wgrad_held"] = None
            ops.wgrad_group(held[0], held[1], defer=self.wgrad_pending)
        if self.wgrad_pending:
            ops.wgrad_reduce(self.wgrad_pending)      # (clears the list in place)

    def join_side(self):
        for side in self.sides:
            torch.cuda.current_stream().wait_stream(side)

    def to_cdt(self, x: torch.Tensor) -> torch.Tensor:
        if self.cdt == torch.float32:
            return x
        out = torch.empty(x.shape, dtype=self.cdt, device=x.device)
        ops.cast_bf16(x.reshape(-1), out.reshape(-1))
        return out


# ------------------------------------------------------------------------------------ encoder
def _encoder_fwd(ctx: StepCtx, idx_all, codes, xs):
    st, L = ctx.store, ctx.store.layout
    T, D = ctx.T, L.D
    if ctx.training and ctx.cdt == torch.bfloat16 and ctx.side is not None:
        # the table gradient's one-hot operand depends on the indices only: build it now, off the
        # critical path, instead of at the tail of the backward pass where nothing overlaps it
        def build():
            ctx.onehot = ops.embed_onehot(idx_all, st.rowoff, L.table_rows_pad)
        ctx.on_side(build, idx_all, which=2)
    h = ops.embed_pool_fwd(idx_all, st.rowoff, st.tables())
    if (_fused_ok(ctx, D) and len(L.num_keys) == 2 and not st.fp8
            and all(x.shape[1] == 512 and x.dtype == torch.bfloat16 for x in xs)):
        # both numerical-attribute Dense layers in one activation-stationary launch (csrc/block_fused.hip)
        return ops.encoder_dense2(xs, [st.cw("encoder/input_%s/kernel" % k) for k in L.num_keys],
                                  [st.weight("encoder/input_%s/bias" % k) for k in L.num_keys], codes, h)
    for j, k in enumerate(L.num_keys):
        width = xs[j].shape[1]
        ops.gemm(xs[j], st.cw("encoder/input_%s/kernel" % k), T, D, width, a_kmajor=True, b_kmajor=True,
                 out=h, accum=True, rowskip=codes[j], bias=st.weight("encoder/input_%s/bias" % k))
    return h


# k-slices of the encoder / heads weight-gradient groups (experiment switches; unset = the library's cost model)
_SPLITK_ENC = int(os.environ.get("MFP_WGRAD_SPLITK_ENC", "0") or 0) or None
_SPLITK_HEADS = int(os.environ.get("MFP_WGRAD_SPLITK_HEADS", "0") or 0) or None


def _encoder_bwd(ctx: StepCtx, idx_all, codes, xs, dh):
    st, L = ctx.store, ctx.store.layout
    T, D = ctx.T, L.D
    # table gradient as a wgrad GEMM (bf16 path; the one-hot kernel counts in LDS words: <= 1024 padded table rows);
    # f32 and larger vocabularies: exact scatter
    onehot = ctx.cdt == torch.bfloat16 and L.table_rows_pad <= 1024
    if L.num_keys or onehot:
        dh_c = ctx.dh_c if (ctx.dh_c is not None and ctx.dh_c.shape == dh.shape) else ctx.to_cdt(dh)
        ctx.dh_c = None
        if onehot and WGRAD_GROUP:
            # the last products of the backward pass -- both encoder Dense gradients (+ biases) and the
            # table gradient -- as ONE grouped launch on the main stream (nothing is left to overlap)
            if ctx.wgrad_pending is None:      # (deferred split-K reductions: ONE flush behind this last group instead)
                ctx.flush_ln_jobs()
            if ctx.onehot is None:
                ctx.onehot = ops.embed_onehot(idx_all, st.rowoff, L.table_rows_pad)
            elif len(ctx.sides) > 2:   # built on side stream 2 during the forward pass
                torch.cuda.current_stream().wait_stream(ctx.sides[2])
            jobs = [dict(A=dh_c, B=xs[j], out=st.grad("encoder/input_%s/kernel" % k), M=D, N=xs[j].shape[1],
                         rowskip=codes[j], colsum=st.grad("encoder/input_%s/bias" % k))
                    for j, k in enumerate(L.num_keys)]
            jobs.append(dict(A=ctx.onehot, B=dh_c, out=st.tables_padded(st.g), M=L.table_rows_pad, N=D))
            ops.wgrad_group(jobs, T, defer=ctx.wgrad_pending, splitk=_SPLITK_ENC)
            ctx.onehot = None
            ctx.join_side()
            ctx.flush_ln_jobs()      # the deferred split-K reductions of the whole backward pass: one launch
            return

        # the last products of the backward pass: nothing on the main stream overlaps them any more,
        # so the independent ones go to different side streams and share the chip
        def wgrad_dense(j, k):
            width = xs[j].shape[1]
            ops.gemm(dh_c, xs[j], D, width, T, a_kmajor=False, b_kmajor=False,
                     out=st.grad("encoder/input_%s/kernel" % k), rowskip_a=codes[j],
                     colsum=st.grad("encoder/input_%s/bias" % k), splitk=ops.wgrad_splitk(T, D, width))

        def wgrad_tables():
            P = ctx.onehot if ctx.onehot is not None else ops.embed_onehot(idx_all, st.rowoff, L.table_rows_pad)
            ctx.onehot = None
            ops.gemm(P, dh_c, L.table_rows_pad, D, T, a_kmajor=False, b_kmajor=False,
                     out=st.tables_padded(st.g), splitk=ops.wgrad_splitk(T, L.table_rows_pad, D))
        if onehot:   # stream 2: the one-hot operand was built there during the forward pass (in order)
            ctx.on_side(wgrad_tables, dh_c, idx_all, which=2)
        n_num = len(L.num_keys)
        for j, k in enumerate(L.num_keys[:-1]):
            ctx.on_side(lambda j=j, k=k: wgrad_dense(j, k), dh_c, xs[j], codes[j], which=j % 2)
        ctx.flush_ln_jobs()   # every LayerNorm layer's parameter-gradient partials: one launch, here in the tail
        if n_num:   # the main stream has nothing else left: the last product runs there, without a fork
            wgrad_dense(n_num - 1, L.num_keys[-1])
    if not onehot:
        ops.embed_pool_bwd(idx_all, st.rowoff, dh, st.tables(st.g))
    ctx.flush_ln_jobs()   # (no-op when already flushed above)
    ctx.join_side()   # last node of the backward pass: every weight gradient is complete after this


class PosConstFn(torch.autograd.Function):
    """``seq += Dropout(PositionEmbedding(range(S)))`` (reference encoder.py:241-242,
    transformer.py:5-30): the learned position token of the non-"set" input types.  An ablation
    path: the existing mask-and-scale kernel (``mfp_dropout_bwd``, stream 0 of the step) applies the
    dropout in both directions, the row gather / batch sum around it is plumbing."""

    POS_RNG_STREAM = 0   # the blocks use streams 2i+1, 2i+2

    @staticmethod
    def forward(fctx, h, anchor, ctx: StepCtx):
        st = ctx.store
        B, S, D = h.shape
        table = st.weight("encoder/input_const/embeddings")[:S]
        fctx.ctx, fctx.shape = ctx, (B, S, D)
        if ctx.p > 0.0:
            tiled = table.repeat(B, 1).contiguous()
            dummy = torch.empty(D, dtype=torch.float32, device=h.device)
            tiled = ops.dropout_bwd(tiled, torch.float32, dummy, ctx.p, ctx.seed, PosConstFn.POS_RNG_STREAM, ctx.step_ptr)
            return h + tiled.view(B, S, D)
        return h + table.unsqueeze(0)

    @staticmethod
    def backward(fctx, dout):
        ctx = fctx.ctx
        B, S, D = fctx.shape
        g = ctx.store.grad("encoder/input_const/embeddings")
        d = dout.contiguous()
        if ctx.p > 0.0:
            dummy = torch.empty(D, dtype=torch.float32, device=d.device)
            d = ops.dropout_bwd(d.view(B * S, D), torch.float32, dummy, ctx.p, ctx.seed, PosConstFn.POS_RNG_STREAM,
                                ctx.step_ptr).view(B, S, D)
        g.zero_()
        g[:S] = d.sum(dim=0)
        return dout, None, None


class ContextTokenFn(torch.autograd.Function):
    """``seq = concat([Embedding(task | length)[:, None], seq], axis=1)`` (reference encoder.py:226-248,
    context in {"id", "length"}).  An ablation path (args.py --context): the row gather / scatter-add
    around the HIP blocks is torch plumbing, like PosConstFn."""

    @staticmethod
    def forward(fctx, h, ids, anchor, ctx: StepCtx):
        table = ctx.store.weight("encoder/input_task/embeddings")
        fctx.ctx, fctx.ids = ctx, ids
        return torch.cat([table[ids][:, None, :], h], dim=1)

    @staticmethod
    def backward(fctx, dout):
        g = fctx.ctx.store.grad("encoder/input_task/embeddings")
        g.zero_()
        g.index_add_(0, fctx.ids, dout[:, 0].contiguous())
        return dout[:, 1:], None, None, None


class EncoderFn(torch.autograd.Function):
    """Encoder on a dict of (already masked) attribute tensors, as the reference's call takes."""

    @staticmethod
    def forward(fctx, anchor, ctx: StepCtx, cat_inputs: List[torch.Tensor], num_inputs: List[torch.Tensor]):
        L = ctx.store.layout
        T = ctx.T
        dev = anchor.device
        n_num = len(L.num_keys)
        cols = [t.reshape(T, -1).to(torch.int32) for t in cat_inputs]
        if n_num:
            cols.append(torch.empty((T, n_num), dtype=torch.int32, device=dev))
        idx_all = torch.cat(cols, dim=1) if len(cols) > 1 else cols[0].contiguous()
        assert idx_all.shape[1] == len(L.idx_cols)
        codes, xs = [], []
        for j, k in enumerate(L.num_keys):
            x = num_inputs[j].reshape(T, -1).contiguous()
            code = torch.empty((T,), dtype=torch.uint8, device=dev)
            ops.row_flags(x, code, idx_all[:, L.special_col[k]:], idx_all.shape[1])
            codes.append(code)
            xs.append(ctx.to_cdt(x))
        fctx.ctx = ctx
        fctx.saved = (idx_all, codes, xs)
        return _encoder_fwd(ctx, idx_all, codes, xs)

    @staticmethod
    def backward(fctx, dh):
        # (RES_GRAD_BF16: dh is a stride-0 placeholder -- the gradient arrived as ctx.dh_c; never materialise it)
        _encoder_bwd(fctx.ctx, *fctx.saved, dh if (dh.dim() > 1 and dh.stride(0) == 0) else dh.contiguous())
        fctx.saved = None
        return None, None, None, None


class EncoderPreFn(torch.autograd.Function):
    """Encoder on the outputs of the fused masking kernel (train-step fast path): the index
    matrix, row codes and compute-dtype numerical rows already exist."""

    @staticmethod
    def forward(fctx, anchor, ctx: StepCtx, idx_all, codes, xs):
        fctx.ctx = ctx
        fctx.saved = (idx_all, codes, xs)
        return _encoder_fwd(ctx, idx_all, codes, xs)

    @staticmethod
    def backward(fctx, dh):
        # (RES_GRAD_BF16: dh is a stride-0 placeholder -- the gradient arrived as ctx.dh_c; never materialise it)
        _encoder_bwd(fctx.ctx, *fctx.saved, dh if (dh.dim() > 1 and dh.stride(0) == 0) else dh.contiguous())
        fctx.saved = None
        return None, None, None, None, None


# -------------------------------------------------------------------------------------- block
class BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(fctx, x, ctx: StepCtx, i: int):
        with ops.scope("block"):
            return BlockFn._forward(fctx, x, ctx, i)

    @staticmethod
    def backward(fctx, dx2):
        with ops.scope("block"):
            return BlockFn._backward(fctx, dx2)

    @staticmethod
    def _forward(fctx, x, ctx: StepCtx, i: int):
        st = ctx.store
        D = st.layout.D
        T, B, S, cdt = ctx.T, ctx.B, ctx.S, ctx.cdt
        p = "blocks/seq2seq_%d/" % i
        x = x.contiguous()
        if (BLOCK_FWD and ATTN_BLOCK and BLOCK_INFER and _fused_ok(ctx, D) and not st.fp8 and _doc_tile_ok(B, S, T)
                and not ctx.training and not fctx.needs_input_grad[0]):
            # inference callers (MFP.__call__(training=False), iterative_decode, eval.py): nothing is saved for a
            # backward pass, so only x1 (re-read as the MLP half's residual) and x2 reach HBM
            fctx.ctx, fctx.i, fctx.saved = ctx, i, None
            return ops.block_infer(
                x, st.weight(p + "norm1/gamma"), st.weight(p + "norm1/beta"), st.cw(p + "attn/dense_query/kernel", rows=3 * D),
                st.span(st.w, p + "attn/dense_query/bias", 3 * D), st.cw(p + "attn/combine_heads/kernel"),
                st.weight(p + "attn/combine_heads/bias"), ctx.nvalid, st.weight(p + "norm2/gamma"), st.weight(p + "norm2/beta"),
                st.cw(p + "mlp/dense_0/kernel"), st.weight(p + "mlp/dense_0/bias"), st.cw(p + "mlp/dense_1/kernel"),
                st.weight(p + "mlp/dense_1/bias"), B, S, NUM_HEADS)
        if BLOCK_FWD and ATTN_BLOCK and _fused_ok(ctx, D) and not st.fp8 and _doc_tile_ok(B, S, T):
            # the whole block in one launch (csrc/block_attn.hip); the last block also leaves the heads' bf16 operand
            x2_c = (torch.empty((T, D), dtype=cdt, device=x.device)
                    if ctx.tail["fuse"] and i == st.layout.L - 1 else None)
            xhat = fctx.needs_input_grad[0] and _xhat_ok(ctx)
            x2, saved = ops.block_fwd(
                x, st.weight(p + "norm1/gamma"), st.weight(p + "norm1/beta"), st.cw(p + "attn/dense_query/kernel", rows=3 * D),
                st.span(st.w, p + "attn/dense_query/bias", 3 * D), st.cw(p + "attn/combine_heads/kernel"),
                st.weight(p + "attn/combine_heads/bias"), ctx.nvalid, st.weight(p + "norm2/gamma"), st.weight(p + "norm2/beta"),
                st.cw(p + "mlp/dense_0/kernel"), st.weight(p + "mlp/dense_0/bias"), st.cw(p + "mlp/dense_1/kernel"),
                st.weight(p + "mlp/dense_1/bias"), B, S, NUM_HEADS, ctx.p, ctx.seed, 2 * i + 1, 2 * i + 2, ctx.step_ptr, x2_c=x2_c,
                xhat_stash=xhat, half_tiles=xhat and _block_half_on(ctx, B, S))
            y1, mean1, rstd1, qkv, a, lse, x1, y2, mean2, rstd2, h = saved      # (xhat: y1 / y2 hold x-hat)
            ctx.tail["x_c"] = (x2, x2_c) if x2_c is not None else None
            fctx.ctx, fctx.i = ctx, i
            fctx.xhat = (xhat, xhat)
            fctx.saved = (x, y1, mean1, rstd1, qkv, a, lse, x1, y2, mean2, rstd2, h)
            return x2
        if _fused512_ok(ctx, D):
            xhat5 = fctx.needs_input_grad[0] and _xhat_ok(ctx)      # (the LN + Dense launches of csrc/block_d512.hip stash x-hat)
            xh1_5 = xh2_5 = False
            # d_model 512: LN1 + Q|K|V | attention | output projection + dropout + residual | LN2 + FFN1 + ReLU | FFN2 + dropout +
            # residual = five launches (csrc/block_d512.hip); the last block also leaves the heads' bf16 operand.  fp8 mode: the
            # two LN + Dense launches are ln_fwd + the MX block-scaled product instead (csrc/gemm_fp8.hip)
            if _fused512_ok(ctx, D, mx_forward=True) or "qkv" not in FP8_PRODUCTS or FP8_WEIGHTS_ONLY:
                wq = (_w8_as_bf16(st, p + "attn/dense_query/kernel", 3 * D) if st.fp8 and FP8_WEIGHTS_ONLY and "qkv" in FP8_PRODUCTS
                      else st.cw(p + "attn/dense_query/kernel", rows=3 * D))
                qkv, y1, mean1, rstd1 = ops.ln_dense_d512(x, st.weight(p + "norm1/gamma"), st.weight(p + "norm1/beta"), wq,
                                                          st.span(st.w, p + "attn/dense_query/bias", 3 * D), 3 * D, xhat_stash=xhat5)
                xh1_5 = xhat5
            else:
                qkv, y1, mean1, rstd1 = _ln_dense(ctx, x, st.weight(p + "norm1/gamma"), st.weight(p + "norm1/beta"),
                                                  st.cw(p + "attn/dense_query/kernel", rows=3 * D), T, 3 * D, D,
                                                  st.span(st.w, p + "attn/dense_query/bias", 3 * D),
                                                  w8=st.w8(p + "attn/dense_query/kernel", 3 * D))
            a, lse = ops.attention_fwd(qkv, ctx.nvalid, B, S, NUM_HEADS)
            x1 = ops.dense_n512_res(a, st.cw(p + "attn/combine_heads/kernel"), st.weight(p + "attn/combine_heads/bias"), x,
                                    (ctx.p, ctx.seed, 2 * i + 1), ctx.step_ptr)
            if _fused512_ok(ctx, D, mx_forward=True) or "ffn1" not in FP8_PRODUCTS or FP8_WEIGHTS_ONLY:
                w1 = (_w8_as_bf16(st, p + "mlp/dense_0/kernel", 2 * D) if st.fp8 and FP8_WEIGHTS_ONLY and "ffn1" in FP8_PRODUCTS
                      else st.cw(p + "mlp/dense_0/kernel"))
                h, y2, mean2, rstd2 = ops.ln_dense_d512(x1, st.weight(p + "norm2/gamma"), st.weight(p + "norm2/beta"), w1,
                                                        st.weight(p + "mlp/dense_0/bias"), 2 * D, relu=True, xhat_stash=xhat5)
                xh2_5 = xhat5
            else:
                h, y2, mean2, rstd2 = _ln_dense(ctx, x1, st.weight(p + "norm2/gamma"), st.weight(p + "norm2/beta"),
                                                st.cw(p + "mlp/dense_0/kernel"), T, 2 * D, D, st.weight(p + "mlp/dense_0/bias"),
                                                relu=True, w8=st.w8(p + "mlp/dense_0/kernel", 2 * D))
            x2_c = (torch.empty((T, D), dtype=cdt, device=x.device)
                    if ctx.tail["fuse"] and i == st.layout.L - 1 else None)
            x2 = ops.dense_n512_res(h, st.cw(p + "mlp/dense_1/kernel"), st.weight(p + "mlp/dense_1/bias"), x1,
                                    (ctx.p, ctx.seed, 2 * i + 2), ctx.step_ptr, out_bf16=x2_c)
            ctx.tail["x_c"] = (x2, x2_c) if x2_c is not None else None
            fctx.ctx, fctx.i = ctx, i
            fctx.xhat = (xh1_5, xh2_5)
            fctx.saved = (x, y1, mean1, rstd1, qkv, a, lse, x1, y2, mean2, rstd2, h)
            return x2
        # (the attention-half-only launch exists for documents of 128 positions; S = 64 is served by the whole-block forms above:
        #  with MFP_BLOCK_FWD=0 such a batch takes the generic launches below)
        attn_half = ATTN_BLOCK and _fused_ok(ctx, D) and not st.fp8 and _doc_tile_ok(B, S, T) and S == 128
        if attn_half:
            # the whole attention half in one launch
            x1, y1, mean1, rstd1, qkv, a, lse = ops.attn_block_fwd(
                x, st.weight(p + "norm1/gamma"), st.weight(p + "norm1/beta"), st.cw(p + "attn/dense_query/kernel", rows=3 * D),
                st.span(st.w, p + "attn/dense_query/bias", 3 * D), st.cw(p + "attn/combine_heads/kernel"),
                st.weight(p + "attn/combine_heads/bias"), ctx.nvalid, B, S, NUM_HEADS, (ctx.p, ctx.seed, 2 * i + 1), ctx.step_ptr)
        elif _fused_ok(ctx, D) and not st.fp8:      # LN1 + Q|K|V in one launch
            qkv, y1, mean1, rstd1 = ops.qkv_fused_fwd(x, st.weight(p + "norm1/gamma"), st.weight(p + "norm1/beta"),
                                                      st.cw(p + "attn/dense_query/kernel", rows=3 * D),
                                                      st.span(st.w, p + "attn/dense_query/bias", 3 * D))
        else:
            qkv, y1, mean1, rstd1 = _ln_dense(ctx, x, st.weight(p + "norm1/gamma"), st.weight(p + "norm1/beta"),
                                          st.cw(p + "attn/dense_query/kernel", rows=3 * D), T, 3 * D, D,
                                          st.span(st.w, p + "attn/dense_query/bias", 3 * D),
                                          w8=st.w8(p + "attn/dense_query/kernel", 3 * D) if st.fp8 else None)
        if not attn_half:
            a, lse = ops.attention_fwd(qkv, ctx.nvalid, B, S, NUM_HEADS)
            x1 = ops.gemm(a, st.cw(p + "attn/combine_heads/kernel"), T, D, D, a_kmajor=True, b_kmajor=True,
                          bias=st.weight(p + "attn/combine_heads/bias"), residual=x,
                          dropout=(ctx.p, ctx.seed, 2 * i + 1), step_ptr=ctx.step_ptr, out_dtype=torch.float32)
        if _fused_ok(ctx, D) and not st.fp8:
            # the last block also leaves the heads' bf16 operand (saves the cast pass in front of the decoder)
            x2_c = (torch.empty((T, D), dtype=cdt, device=x.device)
                    if ctx.tail["fuse"] and i == st.layout.L - 1 else None)
            x2, y2, mean2, rstd2, h = ops.mlp_fused_fwd(
                x1, st.weight(p + "norm2/gamma"), st.weight(p + "norm2/beta"), st.cw(p + "mlp/dense_0/kernel"),
                st.weight(p + "mlp/dense_0/bias"), st.cw(p + "mlp/dense_1/kernel"), st.weight(p + "mlp/dense_1/bias"),
                (ctx.p, ctx.seed, 2 * i + 2), ctx.step_ptr, x2_c=x2_c)
            ctx.tail["x_c"] = (x2, x2_c) if x2_c is not None else None
            fctx.ctx, fctx.i = ctx, i
            fctx.saved = (x, y1, mean1, rstd1, qkv, a, lse, x1, y2, mean2, rstd2, h)
            return x2
        h, y2, mean2, rstd2 = _ln_dense(ctx, x1, st.weight(p + "norm2/gamma"), st.weight(p + "norm2/beta"),
                                        st.cw(p + "mlp/dense_0/kernel"), T, 2 * D, D, st.weight(p + "mlp/dense_0/bias"),
                                        relu=True, w8=st.w8(p + "mlp/dense_0/kernel", 2 * D) if st.fp8 else None)
        x2 = ops.gemm(h, st.cw(p + "mlp/dense_1/kernel"), T, D, 2 * D, a_kmajor=True, b_kmajor=True,
                      bias=st.weight(p + "mlp/dense_1/bias"), residual=x1,
                      dropout=(ctx.p, ctx.seed, 2 * i + 2), step_ptr=ctx.step_ptr, out_dtype=torch.float32)
        fctx.ctx, fctx.i = ctx, i
        fctx.saved = (x, y1, mean1, rstd1, qkv, a, lse, x1, y2, mean2, rstd2, h)
        return x2

    @staticmethod
    def _backward(fctx, dx2):
        ctx, i = fctx.ctx, fctx.i
        st = ctx.store
        D = st.layout.D
        T, B, S, cdt = ctx.T, ctx.B, ctx.S, ctx.cdt
        p = "blocks/seq2seq_%d/" % i
        x, y1, mean1, rstd1, qkv, a, lse, x1, y2, mean2, rstd2, h = fctx.saved
        xhat1, xhat2 = getattr(fctx, "xhat", (False, False))      # y1 / y2 hold x-hat = (x - mean) rstd (mfp_block_fwd_xhat, mfp_ln_dense_d512_xhat)
        r16 = ctx.res_grad is not None      # the residual gradient arrives in bf16 through the context (RES_GRAD_BF16)
        if r16:
            dx2, ctx.res_grad = ctx.res_grad, None
        else:
            dx2 = dx2.contiguous()
        sk = ops.wgrad_splitk
        grouped = WGRAD_GROUP and cdt == torch.bfloat16
        # what the LayerNorm backward reads (the bf16 residual-gradient stream's kernels only)
        xh1, xh2 = (y1 if xhat1 and r16 else None), (y2 if xhat2 and r16 else None)
        na1 = na2 = None
        if xhat1 or xhat2:
            if grouped and ctx.wgrad_pending is not None:
                # x-hat operands: the deferred reduction writes gamma[n] (A^T x-hat)[m][n] + beta[n] colsum[m]  (gamma | beta
                # are neighbours in the flat parameter buffer)
                na1 = st.span(st.w, p + "norm1/gamma", 2 * D) if xhat1 else None
                na2 = st.span(st.w, p + "norm2/gamma", 2 * D) if xhat2 else None
            else:      # (a weight gradient outside the deferred reduction: rebuild the LayerNorm outputs)
                if xhat1:
                    y1 = (y1.float() * st.weight(p + "norm1/gamma") + st.weight(p + "norm1/beta")).to(cdt)
                if xhat2:
                    y2 = (y2.float() * st.weight(p + "norm2/gamma") + st.weight(p + "norm2/beta")).to(cdt)
        # ---- MLP: x2 = x1 + drop(h W2 + b2)
        d_o2 = ctx.handoff.pop(i, None)   # produced by the LN1 backward of block i+1 (fused)
        if d_o2 is None:
            d_o2 = ops.dropout_bwd(dx2, cdt, st.grad(p + "mlp/dense_1/bias"), ctx.p, ctx.seed, 2 * i + 2, ctx.step_ptr)
        wt = st.cwt(p + "mlp/dense_1/kernel")     # [2D][D]: dgrad as a k-major product when kept
        wt0 = st.cwt(p + "mlp/dense_0/kernel")    # [D][2D]
        fused_bwd = _fused_ok(ctx, D) and wt is not None and wt0 is not None
        f512 = _fused512_ok(ctx, D) and wt is not None and wt0 is not None
        ln_fused = False
        if fused_bwd and MLP_BWD_LN and r16 and T % 128 == 0:
            # ... and the backward of LN2 (+ the masked gradient of the attention dropout) in the same launch: dy2 stays on the CU
            dh, dx1, d_o1 = ops.mlp_bwd_ln(d_o2, h, wt, wt0, x1, st.weight(p + "norm2/gamma"), mean2, rstd2, dx2,
                                           st.grad(p + "norm2/gamma"), st.grad(p + "norm2/beta"),
                                           (st.grad(p + "attn/combine_heads/bias"), ctx.p, ctx.seed, 2 * i + 1, ctx.step_ptr),
                                           jobs=ctx.ln_jobs, xhat=xh2, half_tiles=xh2 is not None and _mlp_bwd_half_on(ctx, T))
            ln_fused = True
        elif fused_bwd:      # both input-gradient products of the half in one launch (csrc/block_fused.hip)
            dh, dy2 = ops.mlp_fused_bwd(d_o2, h, wt, wt0)
        elif f512:         # d_model 512 (csrc/block_d512.hip): the ReLU mask in the first product's epilogue, row-owning second
            dh = ops.dense_relumask_d512(d_o2, wt, h)
        else:
            dh = ops.gemm(d_o2, wt if wt is not None else st.cw(p + "mlp/dense_1/kernel"), T, 2 * D, D, a_kmajor=True,
                          b_kmajor=wt is not None, out_dtype=cdt, relu_bwd_aux=h)

        def wgrads_mlp():
            ops.gemm(d_o2, h, D, 2 * D, T, a_kmajor=False, b_kmajor=False, out=st.grad(p + "mlp/dense_1/kernel"),
                     splitk=sk(T, D, 2 * D))
            ops.gemm(dh, y2, 2 * D, D, T, a_kmajor=False, b_kmajor=False, out=st.grad(p + "mlp/dense_0/kernel"),
                     colsum=st.grad(p + "mlp/dense_0/bias"), splitk=sk(T, 2 * D, D))
        if not grouped:
            ctx.on_side(wgrads_mlp, d_o2, h, dh, y2)
        if f512 and D512_LN_BWD and xh2 is not None and T % 128 == 0:
            # LN2 backward (+ the masked gradient of the attention dropout) in the epilogue of dy2 = dh W1 (csrc/block_d512.hip)
            dx1, d_o1 = ops.dense_n512_lnb(dh, wt0, xh2, st.weight(p + "norm2/gamma"), rstd2, dx2,
                                           st.grad(p + "norm2/gamma"), st.grad(p + "norm2/beta"),
                                           drop=(st.grad(p + "attn/combine_heads/bias"), ctx.p, ctx.seed, 2 * i + 1, ctx.step_ptr),
                                           jobs=ctx.ln_jobs)
            ln_fused = True
        elif f512:
            dy2 = ops.dense_n512(dh, wt0)
        elif not fused_bwd:
            dy2 = ops.gemm(dh, wt0 if wt0 is not None else st.cw(p + "mlp/dense_0/kernel"), T, D, 2 * D, a_kmajor=True,
                           b_kmajor=wt0 is not None, out_dtype=cdt)
        # LN2 backward also emits the masked/cast gradient of the attention Dropout + its bias grad
        if not ln_fused:
            dx1, d_o1 = ops.layernorm_bwd(dy2, x1, st.weight(p + "norm2/gamma"), mean2, rstd2, dx2,
                                          st.grad(p + "norm2/gamma"), st.grad(p + "norm2/beta"),
                                          drop=(st.grad(p + "attn/combine_heads/bias"), ctx.p, ctx.seed, 2 * i + 1,
                                                ctx.step_ptr), jobs=ctx.ln_jobs, xhat=xh2)
        # ---- attention: x1 = x + drop(a Wo + bo)
        wt = st.cwt(p + "attn/combine_heads/kernel")
        wtq = st.cwt(p + "attn/dense_query/kernel")    # [D][3D]
        dy1 = None
        ln1_fused = False
        if (_attn_block_bwd_on(ctx) and _fused_ok(ctx, D) and wt is not None and wtq is not None and cdt == torch.bfloat16
                and _doc_tile_ok(B, S, T)):
            if ATTN_BWD_LN and r16:
                # ... and the backward of LN1 (+ the masked gradient block i-1's MLP half starts from) in the same launch
                drop1 = ((st.grad("blocks/seq2seq_%d/mlp/dense_1/bias" % (i - 1)), ctx.p, ctx.seed, 2 * (i - 1) + 2, ctx.step_ptr)
                         if i > 0 else None)
                dqkv, dx, nxt = ops.attn_block_bwd_ln(d_o1, wt, qkv, a, lse, ctx.nvalid, wtq, B, S, NUM_HEADS, x,
                                                      st.weight(p + "norm1/gamma"), mean1, rstd1, dx1,
                                                      st.grad(p + "norm1/gamma"), st.grad(p + "norm1/beta"), drop=drop1,
                                                      jobs=ctx.ln_jobs, xhat=xh1)
                ln1_fused = True
            else:
                # da = d_o1 Wo, attention backward and dy1 = dqkv Wqkv in one launch (csrc/block_attn_bwd.hip)
                dqkv, dy1 = ops.attn_block_bwd(d_o1, wt, qkv, a, lse, ctx.nvalid, wtq, B, S, NUM_HEADS)
        else:
            if _fused_ok(ctx, D) and wt is not None:
                da = ops.dgrad_d256(d_o1, wt)      # activation-stationary (csrc/block_fused.hip)
            elif _fused512_ok(ctx, D) and wt is not None:
                da = ops.dense_n512(d_o1, wt)      # row-owning (csrc/block_d512.hip)
            else:
                da = ops.gemm(d_o1, wt if wt is not None else st.cw(p + "attn/combine_heads/kernel"), T, D, D,
                              a_kmajor=True, b_kmajor=wt is not None, out_dtype=cdt)
            dqkv = ops.attention_bwd(qkv, ctx.nvalid, a, da, lse, B, S, NUM_HEADS)

        def wgrads_attn():
            ops.gemm(d_o1, a, D, D, T, a_kmajor=False, b_kmajor=False, out=st.grad(p + "attn/combine_heads/kernel"),
                     splitk=sk(T, D, D))
            ops.gemm(dqkv, y1, 3 * D, D, T, a_kmajor=False, b_kmajor=False,
                     out=st.span(st.g, p + "attn/dense_query/kernel", 3 * D * D, D),
                     colsum=st.span(st.g, p + "attn/dense_query/bias", 3 * D), splitk=sk(T, 3 * D, D))
        def wgrads_block():   # the four weight gradients (+ two bias gradients) of the block: one launch
            if ctx.wgrad_pending is not None and len(ctx.wgrad_pending) >= ops.WGRAD_MAX_PENDING - 2:
                ops.wgrad_reduce(ctx.wgrad_pending)      # (more than 6 blocks: reduce what has accumulated)
            jobs = [
                dict(A=dqkv, B=y1, out=st.span(st.g, p + "attn/dense_query/kernel", 3 * D * D, D), M=3 * D, N=D,
                     colsum=st.span(st.g, p + "attn/dense_query/bias", 3 * D), naffine=na1),
                dict(A=dh, B=y2, out=st.grad(p + "mlp/dense_0/kernel"), M=2 * D, N=D,
                     colsum=st.grad(p + "mlp/dense_0/bias"), naffine=na2),
                dict(A=d_o2, B=h, out=st.grad(p + "mlp/dense_1/kernel"), M=D, N=2 * D,
                     colsum=st.grad(p + "mlp/dense_1/bias") if i in ctx.tail["bias_wgg"] else None),
                dict(A=d_o1, B=a, out=st.grad(p + "attn/combine_heads/kernel"), M=D, N=D)]
            if _wgrad_pair_on(ctx, D):
                # TWO blocks per grouped launch (round 6): 64 tiles instead of 32 fill the chip with half the k-slices, i.e. half
                # the split-K slab bytes written here and read back by the end-of-backward reduction, and two launches fewer
                # per step.  The operands of the held block stay alive through the job records.
                held = ctx.tail.get("wgrad_held")
                if held is not None:
                    jobs = held[0] + jobs
                if i > 0 and len(jobs) < 4 * min(WGRAD_PAIR, 4):      # (4 products per block; 16 per launch at most)
                    ctx.tail["wgrad_held"] = (jobs, T)
                    return
                ctx.tail["wgrad_held"] = None
            ops.wgrad_group(jobs, T, defer=ctx.wgrad_pending)
        if grouped:
            ctx.on_side(wgrads_block, d_o2, h, dh, y2, d_o1, a, dqkv, y1, hold=True)
        else:
            ctx.on_side(wgrads_attn, d_o1, a, dqkv, y1)
        if dy1 is not None or ln1_fused:
            pass
        elif (_fused_ok(ctx, D) and wtq is not None and DGRAD_LN_HALF and xh1 is not None and T % 64 == 0
              and ops.fused_half_mode(T)):
            # fewer 128-row tiles than CUs (c4): dy1 = dqkv Wqkv on 64-row tiles with the backward of LN1 on its result
            drop1 = ((st.grad("blocks/seq2seq_%d/mlp/dense_1/bias" % (i - 1)), ctx.p, ctx.seed, 2 * (i - 1) + 2, ctx.step_ptr)
                     if i > 0 else None)
            r_ = ops.dgrad_qkv_ln_half(dqkv, wtq, xh1, st.weight(p + "norm1/gamma"), rstd1, dx1,
                                       st.grad(p + "norm1/gamma"), st.grad(p + "norm1/beta"), drop=drop1, jobs=ctx.ln_jobs)
            dx, nxt = r_ if i > 0 else (r_, None)
            ln1_fused = True
        elif _fused_ok(ctx, D) and wtq is not None:
            dy1 = ops.dgrad_qkv(dqkv, wtq)       # activation-stationary (csrc/block_fused.hip)
        elif _fused512_ok(ctx, D) and wtq is not None and D512_LN_BWD and xh1 is not None and T % 128 == 0:
            # LN1 backward (+ the masked gradient block i - 1's MLP half starts from) in the epilogue of dy1 = dqkv Wqkv
            drop1 = ((st.grad("blocks/seq2seq_%d/mlp/dense_1/bias" % (i - 1)), ctx.p, ctx.seed, 2 * (i - 1) + 2, ctx.step_ptr)
                     if i > 0 else None)
            r_ = ops.dense_n512_lnb(dqkv, wtq, xh1, st.weight(p + "norm1/gamma"), rstd1, dx1,
                                    st.grad(p + "norm1/gamma"), st.grad(p + "norm1/beta"), drop=drop1, jobs=ctx.ln_jobs)
            dx, nxt = r_ if i > 0 else (r_, None)
            ln1_fused = True
        elif _fused512_ok(ctx, D) and wtq is not None:
            dy1 = ops.dense_n512(dqkv, wtq)      # row-owning (csrc/block_d512.hip)
        else:
            dy1 = ops.gemm(dqkv, wtq if wtq is not None else st.cw(p + "attn/dense_query/kernel", rows=3 * D), T, D,
                           3 * D, a_kmajor=True, b_kmajor=wtq is not None, out_dtype=cdt)
        if ln1_fused:
            if i > 0:
                ctx.handoff[i - 1] = nxt
            else:
                ctx.dh_c = dx
        elif i > 0:   # dx is the dx2 of block i-1: hand its masked/cast copy over (skips a dropout_bwd)
            pp = "blocks/seq2seq_%d/" % (i - 1)
            dx, nxt = ops.layernorm_bwd(dy1, x, st.weight(p + "norm1/gamma"), mean1, rstd1, dx1,
                                        st.grad(p + "norm1/gamma"), st.grad(p + "norm1/beta"),
                                        drop=(st.grad(pp + "mlp/dense_1/bias"), ctx.p, ctx.seed, 2 * (i - 1) + 2,
                                              ctx.step_ptr), jobs=ctx.ln_jobs, xhat=xh1)
            ctx.handoff[i - 1] = nxt
        elif r16:      # block 0: dx IS the compute-dtype gradient the encoder's weight-gradient products read
            dx = ops.layernorm_bwd(dy1, x, st.weight(p + "norm1/gamma"), mean1, rstd1, dx1,
                                   st.grad(p + "norm1/gamma"), st.grad(p + "norm1/beta"), jobs=ctx.ln_jobs, xhat=xh1)
            ctx.dh_c = dx
        else:
            if cdt == torch.bfloat16:
                # block 0: the fused consumer at rate 0 is a plain compute-dtype copy of dx -- what the
                # encoder's weight-gradient products read (saves the cast pass at the end of the step)
                dx, ctx.dh_c = ops.layernorm_bwd(dy1, x, st.weight(p + "norm1/gamma"), mean1, rstd1, dx1,
                                                 st.grad(p + "norm1/gamma"), st.grad(p + "norm1/beta"),
                                                 drop=(st.scratch("ln_dummy_colsum", (D,), torch.float32), 0.0, 0, 0, None),
                                                 jobs=ctx.ln_jobs)
            else:
                dx = ops.layernorm_bwd(dy1, x, st.weight(p + "norm1/gamma"), mean1, rstd1, dx1,
                                       st.grad(p + "norm1/gamma"), st.grad(p + "norm1/beta"), jobs=ctx.ln_jobs)
        ctx.join_held()
        fctx.saved = None
        if r16:
            if i > 0:
                ctx.res_grad = dx
            return ctx.store.scratch("grad_placeholder", (1,), x.dtype).expand(x.shape), None, None
        return dx, None, None


# ------------------------------------------------------------------------------------ decoder
def _heads_fwd(ctx: StepCtx, h_c: torch.Tensor) -> torch.Tensor:
    st, L = ctx.store, ctx.store.layout
    first = L.head_order[0]
    return ops.gemm(h_c, st.cw("decoder/decoder_%s/kernel" % first, rows=L.Upad), ctx.T, L.Upad, L.D,
                    a_kmajor=True, b_kmajor=True, out_dtype=torch.float32,
                    bias=st.span(st.w, "decoder/decoder_%s/bias" % first, L.Upad))


def _heads_wgrad(ctx: StepCtx, dl_c: torch.Tensor, h_c: torch.Tensor) -> None:
    st, L = ctx.store, ctx.store.layout
    first = L.head_order[0]
    T, D, U = ctx.T, L.D, L.Upad
    def wgrad_heads():
        if WGRAD_GROUP and dl_c.dtype == torch.bfloat16:
            ops.wgrad_group([dict(A=dl_c, B=h_c, out=st.span(st.g, "decoder/decoder_%s/kernel" % first, U * D, D),
                                  M=U, N=D, colsum=st.span(st.g, "decoder/decoder_%s/bias" % first, U))], T,
                            defer=ctx.wgrad_pending, splitk=_SPLITK_HEADS)
            return
        ops.gemm(dl_c, h_c, U, D, T, a_kmajor=False, b_kmajor=False,
                 out=st.span(st.g, "decoder/decoder_%s/kernel" % first, U * D, D),
                 colsum=st.span(st.g, "decoder/decoder_%s/bias" % first, U), splitk=ops.wgrad_splitk(T, U, D))
    ctx.on_side(wgrad_heads, dl_c, h_c)


def _heads_drop(ctx: StepCtx):
    """(p, seed, offset, step_ptr) of the last block's MLP dropout when the heads' input-gradient kernel is to leave the
    masked bf16 gradient that block's backward starts from, else None."""
    L = ctx.store.layout
    if ctx.tail["fuse"] and WGRAD_GROUP and ctx.training and L.L > 0:
        return (ctx.p, ctx.seed, 2 * (L.L - 1) + 2, ctx.step_ptr)
    return None


def _heads_bwd(ctx: StepCtx, dl_c: torch.Tensor, h_c: torch.Tensor, out_dtype=torch.float32) -> torch.Tensor:
    st, L = ctx.store, ctx.store.layout
    first = L.head_order[0]
    T, D, U = ctx.T, L.D, L.Upad
    _heads_wgrad(ctx, dl_c, h_c)
    wt = st.heads_t()
    if MLP_FUSE and wt is not None and dl_c.dtype == torch.bfloat16 and D == 256 and T <= (1 << 19):
        last = L.L - 1
        drop = _heads_drop(ctx)
        if drop is not None:
            # ... and the dropout-masked bf16 copy the last block's backward starts from (mfp_dropout_bwd fused into
            # the epilogue; that Dense's bias gradient then comes out of the block's grouped weight-gradient launch)
            dh, d_o2 = ops.dgrad_rows(dl_c, wt, U, drop=drop)
            ctx.handoff[last] = d_o2
            ctx.tail["bias_wgg"].add(last)
            return dh
        return ops.dgrad_rows(dl_c, wt, U)        # activation-stationary (csrc/block_fused.hip)
    if (D512_FUSE and wt is not None and D == 512 and out_dtype == torch.bfloat16 and dl_c.dtype == torch.bfloat16 and dl_c.is_contiguous()
            and T <= (1 << 19)):
        return ops.dense_n512_lda(dl_c, wt)      # 128 x 256 output tiles (csrc/block_d512.hip) on the transposed, zero-padded heads
    dh = ops.gemm(dl_c, st.cw("decoder/decoder_%s/kernel" % first, rows=U), T, D, U, a_kmajor=True,
                  b_kmajor=False, out_dtype=out_dtype)
    return dh


class DecoderFn(torch.autograd.Function):
    """Heads only: returns the concatenated logits ``[T][Upad]`` (f32)."""

    @staticmethod
    def forward(fctx, h, ctx: StepCtx):
        h_c = ctx.to_cdt(h.contiguous())
        fctx.ctx, fctx.saved = ctx, h_c
        return _heads_fwd(ctx, h_c)

    @staticmethod
    def backward(fctx, dlogits):
        ctx = fctx.ctx
        dh = _heads_bwd(ctx, ctx.to_cdt(dlogits.contiguous()), fctx.saved)
        fctx.saved = None
        return dh, None


def loss_row_maps(sort, logits, nvalid, B, S):
    """(pred_row, true_row) of the position-sorted loss (reference metrics.py:180-211): targets are
    ordered by their labels, predictions by their own argmax, both only for the flagged documents."""
    if sort is None:
        return None, None
    pred_row = true_row = None
    if sort.get("ignore_sort") != "gt":
        true_row = ops.sort_positions(nvalid, sort["flag"], B, S, labels=sort["labels"])
    if sort.get("ignore_sort") != "pred":
        pred_row = ops.sort_positions(nvalid, sort["flag"], B, S, logits=logits, heads=sort["heads"])
    return pred_row, true_row


class DecoderLossFn(torch.autograd.Function):
    """Heads + fused LossLayer.  Returns ``(loss_total, sums[nkeys,3], logits)``; the backward
    assumes the conventional unit upstream gradient on ``loss_total``."""

    @staticmethod
    def forward(fctx, h, ctx: StepCtx, keys: List[dict]):
        # the unused output gradients (sums, logits) must NOT be materialised: autograd would
        # zero-fill a [T, Upad] f32 tensor (181 MB at the Crello config) every step
        fctx.set_materialize_grads(False)
        h = h.contiguous()
        xc = ctx.tail["x_c"]
        if xc is not None and xc[0].data_ptr() == h.data_ptr() and xc[1].shape == h.shape:
            h_c = xc[1]           # written by the last block's MLP kernel
        else:
            h_c = ctx.to_cdt(h)
        ctx.tail["x_c"] = None
        st, L = ctx.store, ctx.store.layout
        if (HEADS_FUSED and ctx.cdt == torch.bfloat16 and ctx.loss_sort is None and MLP_FUSE
                and ops.heads_loss_fused_ok(keys, L.Upad, L.D, ctx.T, ctx.tail.get("want_logits", True))):
            # heads + losses + d(loss)/d(h) in ONE launch (csrc/heads_loss.hip): the f32 logits are written only when
            # somebody wants them, the per-key sums come back as per-workgroup partials
            first = L.head_order[0]
            dl = st.scratch("dlogits", (ctx.T, L.Upad), ctx.cdt)
            drop = _heads_drop(ctx)
            r16 = drop is not None and fctx.needs_input_grad[0] and _res16_ok(ctx)
            part, dl, logits, dx, dxd = ops.heads_loss_fused(
                h_c, st.cw("decoder/decoder_%s/kernel" % first, rows=L.Upad),
                st.span(st.w, "decoder/decoder_%s/bias" % first, L.Upad), keys, ctx.nvalid, ctx.B, ctx.S, dlogits=dl,
                want_logits=ctx.tail.get("want_logits", True), drop=drop,
                dx_dtype=torch.bfloat16 if r16 else torch.float32)
            n3 = 3 * len(keys)
            flat = ctx.tail["sums"]
            if flat is not None and flat.numel() == n3 + 1:
                sums, loss = flat[:n3].view(len(keys), 3), flat[n3:].view(())
                ctx.tail["sums"] = None
                if ctx.ln_jobs is not None and fctx.needs_input_grad[0]:      # (a backward pass will follow)
                    # summed with the LayerNorm parameter-gradient partials, in the one launch at the end of the backward pass
                    ctx.ln_jobs.append(dict(part=part, out0=flat, out1=None, out2=None, split1=n3, split2=n3,
                                            P=part.shape[0], N=n3, pstride=part.shape[1]))
                else:
                    ops.reduce_partials(part, flat, n3)
            else:
                sums = torch.empty((len(keys), 3), dtype=torch.float32, device=h_c.device)
                ops.reduce_partials(part, sums, n3)
                loss = sums[:, 0].sum()
            if logits is None:
                logits = h_c.new_empty((0, L.Upad), dtype=torch.float32)
            fctx.ctx, fctx.saved = ctx, (h_c, dl, dx, dxd)
            fctx.hshape = (h.shape, h.dtype)
            fctx.mark_non_differentiable(sums, logits)
            return loss, sums, logits
        logits = _heads_fwd(ctx, h_c)
        dl = ctx.store.scratch("dlogits", logits.shape, ctx.cdt)   # zeroed once: pad columns stay 0
        pred_row, true_row = loss_row_maps(ctx.loss_sort, logits, ctx.nvalid, ctx.B, ctx.S)
        flat = ctx.tail["sums"]
        if flat is not None and flat.numel() == 3 * len(keys) + 1:
            # the train step (MFP._forward): accumulators zeroed by the step prologue -> no zeroing launch; and the
            # returned "loss" is only the ROOT of the backward pass (a zero scalar): nothing on the device needs its
            # value -- the step's loss is the host-side sum of sums[:, 0] (MFP.metrics_dict) -- and both a reduction
            # launch (4.8 us) and a grand-total atomic in the loss kernels (+33 us: 5 k same-address atomics) cost more
            sums = flat[:3 * len(keys)].view(len(keys), 3)
            ops.loss_fwd_bwd(logits, keys, ctx.nvalid, ctx.B, ctx.S, ctx.cdt, sums=sums, dlogits=dl,
                             pred_row=pred_row, true_row=true_row, prezeroed=True)
            loss = flat[3 * len(keys):].view(())
            ctx.tail["sums"] = None
        else:
            sums, dl = ops.loss_fwd_bwd(logits, keys, ctx.nvalid, ctx.B, ctx.S, ctx.cdt, dlogits=dl,
                                        pred_row=pred_row, true_row=true_row)
            loss = sums[:, 0].sum()
        fctx.ctx, fctx.saved = ctx, (h_c, dl)
        fctx.mark_non_differentiable(sums, logits)
        return loss, sums, logits

    @staticmethod
    def backward(fctx, dloss, dsums, dlogits):
        ctx = fctx.ctx
        if len(fctx.saved) == 4:      # the one-launch path: the input gradient came out of the forward launch
            h_c, dl, dh, dxd = fctx.saved
            _heads_wgrad(ctx, dl, h_c)
            if dxd is not None:
                last = ctx.store.layout.L - 1
                ctx.handoff[last] = dxd
                ctx.tail["bias_wgg"].add(last)
            fctx.saved = None
            if dh.dtype == torch.bfloat16:      # RES_GRAD_BF16: the gradient goes down in StepCtx, autograd gets a placeholder
                ctx.res_grad = dh
                return ctx.store.scratch("grad_placeholder", (1,), fctx.hshape[1]).expand(fctx.hshape[0]), None, None
            return dh, None, None
        h_c, dl = fctx.saved
        r16 = fctx.needs_input_grad[0] and _res16_ok(ctx) and ctx.store.layout.D != 256
        dh = _heads_bwd(ctx, dl, h_c, out_dtype=torch.bfloat16 if r16 else torch.float32)
        fctx.saved = None
        if r16:      # RES_GRAD_BF16 at d_model 512: the plain product writes the compute dtype, the gradient goes down in StepCtx
            ctx.res_grad = dh
            return ctx.store.scratch("grad_placeholder", (1,), torch.float32).expand(h_c.shape), None, None
        return dh, None, None
