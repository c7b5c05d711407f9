"""Tensor-level wrappers over the C-ABI (no autograd here; see ``mfp.hip.functions``).

PyTorch-ROCm is plumbing only: it owns device memory (caching allocator) and the stream; every
wrapper passes ``tensor.data_ptr()`` + ``torch.cuda.current_stream().cuda_stream`` to the
library and never computes anything itself.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import (GEMM_ACCUM, GEMM_BIAS, GEMM_COLSUM_A, GEMM_DROPOUT, GEMM_RELU, GEMM_RELU_BWD,
               GEMM_RESIDUAL, GEMM_ROWSKIP, GEMM_ROWSKIP_A, MFP_BF16, MFP_F32, GemmArgs, LossKey,
               MaskCol, WgradJob, WgradPending, check, load)

_DT = {torch.float32: MFP_F32, torch.bfloat16: MFP_BF16}
LN_EPS = 1e-3  # Keras LayerNormalization() default [TF-EXT]


def dt_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError("unsupported compute dtype %s (float32 or bfloat16)" % dtype) from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libmfp_hip ops need HIP device tensors (got %s); there is no CPU "
                           "fallback on the product path" % t.device)
    return t.data_ptr()


# ------------------------------------------------------------------------------- profiling
# bench.py's roofline leg: HIP events recorded on the launch stream around each library call
# (torch.cuda.Event records on torch's current stream, which is the stream passed to the kernels).
_prof = None
_scope = [""]


class scope:
    """Tag the library calls made inside the ``with`` block (bench.py sums the kernels of the
    encoder blocks separately: the quantity north_star's MFMA target is stated on)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        _scope.append(self.name)

    def __exit__(self, *exc):
        _scope.pop()


def start_profile():
    global _prof
    _prof = []


def stop_profile():
    """-> [(kernel name, algorithmic flops, algorithmic bytes, milliseconds, scope, share of one launch)]"""
    global _prof
    recs, _prof = _prof or [], None
    torch.cuda.synchronize()
    return [(n, f, b, e0.elapsed_time(e1) * frac, sc, frac) for n, f, b, sc, e0, e1, frac in recs]


class _timed:
    """``split`` = [(scope, flops, nbytes)]: one launch that serves several scopes (the merged weight-gradient launch
    and its reduction: blocks + heads + encoder) is booked once per scope, its time divided in proportion to the flops
    (bytes when there are none)."""

    def __init__(self, name, flops=0, nbytes=0, split=None):
        self.args = (name, flops, nbytes, _scope[-1])
        self.split = split

    def __enter__(self):
        if _prof is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if _prof is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            if self.split:
                merged = {}
                for sc, f, b in self.split:
                    m = merged.setdefault(sc, [0, 0])
                    m[0] += f
                    m[1] += b
                tot = sum(m[0] for m in merged.values()) or sum(m[1] for m in merged.values()) or 1
                use = 0 if any(m[0] for m in merged.values()) else 1
                for sc, m in merged.items():
                    _prof.append((self.args[0], m[0], m[1], sc, self.e0, e1, m[use] / tot))
            else:
                _prof.append(self.args + (self.e0, e1, 1.0))


def _esz(t):
    return t.element_size()


# ------------------------------------------------------------------------------- workspace
_workspaces: Dict[tuple, List[torch.Tensor]] = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream): reuse is stream-ordered, so kernels running
    concurrently on the side stream (weight gradients) never share partial buffers with the main
    stream; old buffers are kept alive so pointers baked into a captured hipGraph stay valid."""
    device = torch.device(device)
    lst = _workspaces.setdefault((device, torch.cuda.current_stream().cuda_stream), [])
    if not lst or lst[-1].numel() < nbytes:
        lst.append(torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device))
    return lst[-1]


# ------------------------------------------------------------------------------------ GEMM
def gemm(A: torch.Tensor, B: torch.Tensor, M: int, N: int, K: int, *, a_kmajor: bool, b_kmajor: bool,
         out: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None,
         bias: Optional[torch.Tensor] = None, relu: bool = False,
         residual: Optional[torch.Tensor] = None, dropout: Optional[Tuple[float, int, int]] = None,
         accum: bool = False, rowskip: Optional[torch.Tensor] = None,
         relu_bwd_aux: Optional[torch.Tensor] = None, colsum: Optional[torch.Tensor] = None,
         rowskip_a: Optional[torch.Tensor] = None, splitk: int = 1,
         step_ptr: Optional[torch.Tensor] = None,
         lda: Optional[int] = None, ldb: Optional[int] = None, ldc: Optional[int] = None) -> torch.Tensor:
    """``out[M,N] = epilogue(op(A) @ op(B))`` -- see ``mfp_gemm`` in include/mfp_hip.h."""
    lib = load()
    assert A.dtype == B.dtype, (A.dtype, B.dtype)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or A.dtype, device=A.device)
    a = GemmArgs()
    a.A, a.B, a.C = _ptr(A), _ptr(B), _ptr(out)
    a.M, a.N, a.K = M, N, K
    a.lda = lda if lda is not None else (K if a_kmajor else M)
    a.ldb = ldb if ldb is not None else (K if b_kmajor else N)
    a.ldc = ldc if ldc is not None else N
    a.a_kmajor, a.b_kmajor = int(a_kmajor), int(b_kmajor)
    a.in_dtype, a.out_dtype = dt_code(B.dtype), dt_code(out.dtype)
    flags = 0
    if bias is not None:
        flags |= GEMM_BIAS
        a.bias = _ptr(bias)
    if relu:
        flags |= GEMM_RELU
    if residual is not None:
        flags |= GEMM_RESIDUAL
        a.residual = _ptr(residual)
    if dropout is not None and dropout[0] > 0.0:
        flags |= GEMM_DROPOUT
        a.dropout_p, a.seed, a.offset = float(dropout[0]), int(dropout[1]), int(dropout[2])
        a.step_ptr = _ptr(step_ptr)
    if accum:
        flags |= GEMM_ACCUM
    if rowskip is not None:
        flags |= GEMM_ROWSKIP
        a.rowcode = _ptr(rowskip)
    if relu_bwd_aux is not None:
        flags |= GEMM_RELU_BWD
        a.aux = _ptr(relu_bwd_aux)
    if colsum is not None:
        flags |= GEMM_COLSUM_A
        a.colsum = _ptr(colsum)
    if rowskip_a is not None:
        flags |= GEMM_ROWSKIP_A
        a.rowcode = _ptr(rowskip_a)
    a.flags, a.splitk = flags, splitk
    nbytes = lib.mfp_gemm_workspace_bytes(ctypes.byref(a))
    if nbytes:
        ws = workspace(nbytes, A.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    name = "gemm_kernel"
    if _prof is not None:   # booked under the kernel family the library actually launches
        name = lib.mfp_gemm_kernel_family(ctypes.byref(a)).decode()
        if name == "gemm_kernel":
            name = "gemm_kernel<%s,%s,%s>" % ("bf16" if A.dtype == torch.bfloat16 else "f32", "Ak" if a_kmajor else "Am",
                                              "Bk" if b_kmajor else "Bn")
    # algorithmic bytes: both operands once, the result once, plus the epilogue operands it reads
    nbytes = M * K * _esz(A) + N * K * _esz(B) + M * N * _esz(out)
    nbytes += M * N * 4 * ((residual is not None) + bool(accum)) + (M * N * _esz(relu_bwd_aux) if relu_bwd_aux is not None else 0)
    with _timed(name, 2 * M * N * K, nbytes):
        check(lib.mfp_gemm(ctypes.byref(a), _stream()), "mfp_gemm")
    return out


_cu_count: Dict = {}


_reserved_cus = 0


def set_reserved_cus(n: int) -> None:
    """Leave ``n`` CUs free in every persistent launch (see mfp_set_reserved_cus): room for RCCL's workgroups when the
    all-reduce of a gradient bucket overlaps the backward pass (``MFP_DP_RESERVE_CUS``, mfp/dp.py)."""
    global _reserved_cus
    check(load().mfp_set_reserved_cus(int(n)), "mfp_set_reserved_cus")
    _reserved_cus = int(n)


def cu_count(device=None) -> int:
    """Compute units of ``device`` (the current HIP device by default): what a persistent launch aims its grid at."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev not in _cu_count:
        _cu_count[dev] = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    return _cu_count[dev]


def fused_half_mode(T: int) -> bool:
    """What csrc/block_fused.hip's half_mode() decides for the activation-stationary launches of T rows: the half-size (64-row)
    workgroups when MFP_FUSED_HALF says so, else when the 128-row grid would leave CUs idle."""
    env = os.environ.get("MFP_FUSED_HALF", "")
    if env != "":
        return int(env) != 0
    return (T + 127) // 128 < cu_count()


def wgrad_splitk(T: int, M: int, N: int) -> int:
    """Split the token contraction so that the 128x128 output tiles x splits give about one
    persistent workgroup per CU (the streaming wgrad kernel keeps 8 waves and ~80 KB of LDS per
    workgroup: one per CU), in multiples of 8 (a k-chunk's tiles then share an XCD)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    sk = max(1, (cu_count() - _reserved_cus) // max(tiles, 1))
    if sk >= 8:
        sk = sk // 8 * 8
    while sk > 1 and T // sk < 256:
        sk //= 2
    return sk


# ------------------------------------------------------------------------ fp8 forward Dense (MX block-scaled)
def quantize_mxfp8(w: torch.Tensor, rows: int, K: int, out: torch.Tensor, scales: torch.Tensor) -> None:
    """w f32 [rows * K] -> out (uint8) e4m3 elements, scales (uint8 [rows * K / 32]) e8m0 block scales (OCP MX, block 32)."""
    lib = load()
    check(lib.mfp_quantize_mxfp8(_ptr(w), rows, K, _ptr(out), _ptr(scales), _stream()), "mfp_quantize_mxfp8")


def gemm_mxfp8(X: torch.Tensor, Wq: torch.Tensor, Ws: torch.Tensor, M: int, N: int, K: int, bias: Optional[torch.Tensor] = None,
               relu: bool = False) -> torch.Tensor:
    """bf16 [M, N] = relu?(X @ W^T + bias) as an MX product: X bf16 quantised on the fly (e4m3 elements, e8m0 scale per
    32 k), Wq [N, K] / Ws [N, K / 32] from quantize_mxfp8."""
    lib = load()
    assert X.dtype == torch.bfloat16 and X.stride(1) == 1 and Wq.dtype == torch.uint8 and Ws.dtype == torch.uint8
    out = torch.empty((M, N), dtype=torch.bfloat16, device=X.device)
    with _timed("gemm_mxfp8_kernel", 2 * M * N * K, M * K * 2 + N * K + M * N * 2):
        check(lib.mfp_gemm_mxfp8(_ptr(X), _ptr(Wq), _ptr(Ws), _ptr(bias), _ptr(out), M, N, K, X.stride(0), N, int(relu),
                                 _stream()), "mfp_gemm_mxfp8")
    return out


# ---------------------------------------------------------------- grouped weight gradients
_tickets: Dict[tuple, torch.Tensor] = {}
WGRAD_MAX_TILES = 4096


def _wgrad_tickets(device) -> torch.Tensor:
    """Per (device, stream) ticket words of the in-launch split-K reduction: zero before the first
    launch, left zero by every launch; launches on different streams may overlap, so each stream has
    its own (like the workspaces)."""
    key = (torch.device(device), torch.cuda.current_stream().cuda_stream)
    t = _tickets.get(key)
    if t is None:
        t = _tickets[key] = torch.zeros(WGRAD_MAX_TILES, dtype=torch.int32, device=device)
    return t


_deferred_ws: Dict = {}


def _deferred_workspace(nbytes: int, device, slot: int) -> torch.Tensor:
    """Slab buffer number ``slot`` of a step's deferred weight-gradient groups: each pending group keeps its partial
    tiles until the reduction launch, so (unlike ``workspace``) the groups of one step must not share a buffer.
    Grow-only per (device, stream, slot); old buffers stay alive for captured graphs."""
    key = (torch.device(device), torch.cuda.current_stream().cuda_stream, slot)
    lst = _deferred_ws.setdefault(key, [])
    if not lst or lst[-1].numel() < nbytes:
        lst.append(torch.empty(int(nbytes), dtype=torch.uint8, device=device))
    return lst[-1]


WGRAD_MAX_PENDING = 8
WGRAD_MAX_PENDING_JOBS = 56      # csrc/gemm_wgg.h: WGR_MAX_JOBS


def _wgrad_jobs_array(jobs: Sequence[dict]):
    arr = (WgradJob * len(jobs))()
    for i, j in enumerate(jobs):
        a = arr[i]
        a.A, a.B, a.C = _ptr(j["A"]), _ptr(j["B"]), _ptr(j["out"])
        a.colsum, a.rowcode = _ptr(j.get("colsum")), _ptr(j.get("rowskip"))
        a.M, a.N = j["M"], j["N"]
        a.lda, a.ldb = j["A"].stride(0), j["B"].stride(0)
        a.ldc = j["out"].stride(0) if j["out"].dim() == 2 else j["N"]
    return arr


def wgrad_group_splitk(jobs: Sequence[dict], K: int, deferred: bool = True) -> int:
    """The token split ``mfp_wgrad_group_splitk`` picks for this group on this device (``deferred``: for the partial-tile
    launch of the train step, else for the launch that reduces in place)."""
    return int(load().mfp_wgrad_group_splitk(_wgrad_jobs_array(jobs), len(jobs), K, int(deferred)))


def wgrad_group(jobs: Sequence[dict], K: int, splitk: Optional[int] = None, defer: Optional[list] = None) -> None:
    """``out_j[M_j, N_j] = A_j[K, M_j]^T @ B_j[K, N_j]`` (+ ``colsum_j[M_j] = sum_k A_j``) for up to 8 jobs
    in ONE launch -- see ``mfp_wgrad_group`` in include/mfp_hip.h.  jobs: dicts with A, B (bf16
    [K, ld]), out (f32 [M, ldc] view), M, N and optionally colsum (f32 [M]), rowskip (u8 [K]).

    ``naffine`` (f32 [2 N] = gamma | beta, deferred form only): B holds x-hat = (x - mean) rstd of the LayerNorm whose output
    is the operand meant; the reduction writes gamma[n] (A^T x-hat)[m][n] + beta[n] colsum[m].

    ``defer`` (a list): the launch only leaves its split-K partial tiles (``mfp_wgrad_group_partial``) and appends a
    record to the list; ``wgrad_reduce(defer)`` later writes the gradients of every recorded group in one launch."""
    lib = load()
    n = len(jobs)
    arr = (WgradJob * n)()
    flops = nbytes = 0
    for i, j in enumerate(jobs):
        A, B, out = j["A"], j["B"], j["out"]
        assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and out.dtype == torch.float32
        assert A.shape[0] == K and B.shape[0] == K and A.stride(1) == 1 and B.stride(1) == 1 and out.stride(-1) == 1
        a = arr[i]
        a.A, a.B, a.C = _ptr(A), _ptr(B), _ptr(out)
        a.colsum, a.rowcode = _ptr(j.get("colsum")), _ptr(j.get("rowskip"))
        a.M, a.N = j["M"], j["N"]
        a.lda, a.ldb = A.stride(0), B.stride(0)
        a.ldc = out.stride(0) if out.dim() == 2 else j["N"]
        na = j.get("naffine")      # f32 [2 N] = gamma | beta: B holds x-hat of that LayerNorm (deferred form only)
        if na is not None:
            assert defer is not None and j.get("colsum") is not None and na.dtype == torch.float32 and na.numel() == 2 * a.N and na.is_contiguous()
        a.n_affine = _ptr(na)
        flops += 2 * K * a.M * a.N
        nbytes += K * (a.M + a.N) * 2 + a.M * a.N * 4
    dev = jobs[0]["A"].device
    if splitk is None:
        splitk = lib.mfp_wgrad_group_splitk(arr, n, K, int(defer is not None))
    assert lib.mfp_wgrad_group_tiles(arr, n) <= WGRAD_MAX_TILES
    need = lib.mfp_wgrad_group_workspace_bytes(arr, n, splitk)
    if defer is not None:
        if sum(r["n"] for r in defer) + n > WGRAD_MAX_PENDING_JOBS:      # (mfp_wgrad_reduce takes 56 jobs per launch)
            wgrad_reduce(defer)
        assert len(defer) < WGRAD_MAX_PENDING, "flush the pending weight-gradient groups first (wgrad_reduce)"
        ws = _deferred_workspace(need, dev, len(defer))
        with _timed("gemm_wgg_kernel", flops, nbytes):
            check(lib.mfp_wgrad_group_partial(arr, n, K, splitk, ws.data_ptr(), ws.numel(), _stream()), "mfp_wgrad_group_partial")
        defer.append(dict(arr=arr, n=n, splitk=splitk, ws=ws, scope=_scope[-1],
                          # (the partial kernel has consumed A / B / rowskip in stream order: only what the reduction
                          #  launch writes is kept alive -- views of the persistent gradient buffer)
                          keep=[(j["out"], j.get("colsum")) for j in jobs],
                          nbytes=need + sum(j["M"] * j["N"] * 4 for j in jobs)))
        return
    ws = workspace(need, dev)
    with _timed("gemm_wgg_kernel", flops, nbytes):
        check(lib.mfp_wgrad_group(arr, n, K, splitk, ws.data_ptr(), ws.numel(), _wgrad_tickets(dev).data_ptr(),
                                  _stream()), "mfp_wgrad_group")


def wgrad_reduce(pending: list) -> None:
    """The gradients of every group ``wgrad_group(..., defer=pending)`` recorded: ONE launch (``mfp_wgrad_reduce``) sums each
    group's split-K slabs in the fixed order of the in-launch reduction.  Empties the list."""
    if not pending:
        return
    lib = load()
    n = len(pending)
    arr = (WgradPending * n)()
    nbytes = 0
    for i, rec in enumerate(pending):
        arr[i].jobs, arr[i].njobs, arr[i].splitk, arr[i].workspace = rec["arr"], rec["n"], rec["splitk"], rec["ws"].data_ptr()
        nbytes += rec["nbytes"]
    with _timed("wgg_reduce_kernel", split=[(rec["scope"], 0, rec["nbytes"]) for rec in pending]):
        check(lib.mfp_wgrad_reduce(arr, n, _stream()), "mfp_wgrad_reduce")
    pending.clear()


# ------------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out_dtype: torch.dtype):
    lib = load()
    T, D = x.shape
    y = torch.empty((T, D), dtype=out_dtype, device=x.device)
    mean = torch.empty((T,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((T,), dtype=torch.float32, device=x.device)
    with _timed("ln_fwd_kernel", 0, T * D * (4 + _esz(y))):
        check(lib.mfp_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), T, D,
                                    LN_EPS, dt_code(out_dtype), _stream()), "mfp_layernorm_fwd")
    return y, mean, rstd


def mlp_fused_fwd(x1, gamma, beta, W1, b1, W2, b2, dropout: Tuple[float, int, int], step_ptr=None, x2_c=None):
    """x2 = x1 + Dropout(relu(LN(x1) W1^T + b1) W2^T + b2) in one launch (d_model 256, bf16 weights).
    Returns (x2, y2, mean, rstd, h) -- the tensors the three-launch path saves for backward.
    ``x2_c`` (bf16 [T, D], optional): also receives a bf16 copy of x2."""
    lib = load()
    T, D = x1.shape
    F = 2 * D
    dev = x1.device
    y2 = torch.empty((T, D), dtype=torch.bfloat16, device=dev)
    h = torch.empty((T, F), dtype=torch.bfloat16, device=dev)
    mean = torch.empty((T,), dtype=torch.float32, device=dev)
    rstd = torch.empty((T,), dtype=torch.float32, device=dev)
    x2 = torch.empty((T, D), dtype=torch.float32, device=dev)
    with _timed("mlp_fused_kernel", 2 * 2 * T * D * F, T * (D * 4 * 3 + D * 2 + F * 2) + 2 * D * F * 2):
        check(lib.mfp_mlp_fused_fwd(_ptr(x1), _ptr(gamma), _ptr(beta), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2),
                                    _ptr(y2), _ptr(mean), _ptr(rstd), _ptr(h), _ptr(x2), _ptr(x2_c), T, D, LN_EPS,
                                    float(dropout[0]), int(dropout[1]), int(dropout[2]),
                                    _ptr(step_ptr) if step_ptr is not None else None, _stream()),
              "mfp_mlp_fused_fwd")
    return x2, y2, mean, rstd, h


def attn_block_fwd(x, gamma, beta, Wqkv, bqkv, Wo, bo, nvalid, B: int, S: int, H: int, dropout: Tuple[float, int, int],
                   step_ptr=None):
    """x1 = x + Dropout(MHSA(LN(x)) Wo^T + bo) in ONE launch (d_model 256, S = 128, 8 heads; see mfp_attn_block_fwd).
    Returns (x1, y1, mean, rstd, qkv, a, lse) -- what the three launches it replaces save for the backward pass."""
    lib = load()
    T, D = x.shape
    dev = x.device
    y1 = torch.empty((T, D), dtype=torch.bfloat16, device=dev)
    qkv = torch.empty((T, 3 * D), dtype=torch.bfloat16, device=dev)
    a = torch.empty((T, D), dtype=torch.bfloat16, device=dev)
    mean = torch.empty((T,), dtype=torch.float32, device=dev)
    rstd = torch.empty((T,), dtype=torch.float32, device=dev)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    x1 = torch.empty((T, D), dtype=torch.float32, device=dev)
    flops = 2 * T * D * 3 * D + 4 * B * S * S * D + 2 * T * D * D
    with _timed("attn_block_fwd_kernel", flops, T * (D * 4 * 3 + D * 2 * 2 + 3 * D * 2) + 4 * D * D * 2):
        check(lib.mfp_attn_block_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(Wqkv), _ptr(bqkv), _ptr(Wo), _ptr(bo), _ptr(nvalid),
                                     _ptr(y1), _ptr(mean), _ptr(rstd), _ptr(qkv), _ptr(a), _ptr(lse), _ptr(x1), B, S, D, H, LN_EPS,
                                     float(dropout[0]), int(dropout[1]), int(dropout[2]),
                                     _ptr(step_ptr) if step_ptr is not None else None, _stream()), "mfp_attn_block_fwd")
    return x1, y1, mean, rstd, qkv, a, lse


# waves per half-document workgroup of mfp_block_fwd_xhat_half (4: two row tiles per wave; 8: one) -- A/B switch
HALF_WAVES = int(os.environ.get("MFP_BLOCK_HALF_WAVES", "8"))


def block_fwd(x, gamma, beta, Wqkv, bqkv, Wo, bo, nvalid, gamma2, beta2, W1, b1, W2, b2, B: int, S: int, H: int,
              p: float, seed: int, off_attn: int, off_mlp: int, step_ptr=None, x2_c=None, xhat_stash: bool = False,
              half_tiles: bool = False):
    """A whole DeepSVG block forward in ONE launch (see mfp_block_fwd): returns
    (x2, (y1, mean1, rstd1, qkv, a, lse, x1, y2, mean2, rstd2, h)) -- the tensors the separate launches save.
    ``xhat_stash``: y1 / y2 hold x-hat = (x - mean) rstd instead of the LayerNorm outputs (mfp_block_fwd_xhat);
    ``half_tiles`` (with xhat_stash, S = 128): two workgroups per document (mfp_block_fwd_xhat_half), bit-identical results;
    True = MFP_BLOCK_HALF_WAVES waves per workgroup, 4 / 8 = that many."""
    if half_tiles and S == 64:
        half_tiles = 8      # (a half tile is one document of 64 positions: the eight-wave form only)
    assert not half_tiles or (xhat_stash and S in (64, 128))
    lib = load()
    T, D = x.shape
    dev = x.device
    bf, f32 = torch.bfloat16, torch.float32
    y1, y2, a = (torch.empty((T, D), dtype=bf, device=dev) for _ in range(3))
    qkv = torch.empty((T, 3 * D), dtype=bf, device=dev)
    h = torch.empty((T, 2 * D), dtype=bf, device=dev)
    mean1, rstd1, mean2, rstd2 = (torch.empty((T,), dtype=f32, device=dev) for _ in range(4))
    lse = torch.empty((B, H, S), dtype=f32, device=dev)
    x1, x2 = (torch.empty((T, D), dtype=f32, device=dev) for _ in range(2))
    flops = 2 * T * D * 3 * D + 4 * B * S * S * D + 2 * T * D * D + 2 * 2 * T * D * 2 * D
    nbytes = T * (D * 4 * 5 + D * 2 * 3 + 3 * D * 2 + 2 * D * 2) + 8 * D * D * 2
    with _timed("block_fwd_kernel", flops, nbytes):
        args = (_ptr(x), _ptr(gamma), _ptr(beta), _ptr(Wqkv), _ptr(bqkv), _ptr(Wo), _ptr(bo), _ptr(nvalid),
                _ptr(y1), _ptr(mean1), _ptr(rstd1), _ptr(qkv), _ptr(a), _ptr(lse), _ptr(x1), _ptr(gamma2), _ptr(beta2),
                _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(y2), _ptr(mean2), _ptr(rstd2), _ptr(h), _ptr(x2), _ptr(x2_c),
                B, S, D, H, LN_EPS, float(p), int(seed), int(off_attn), int(off_mlp),
                _ptr(step_ptr) if step_ptr is not None else None)
        if half_tiles:
            check(lib.mfp_block_fwd_xhat_half(*args, int(half_tiles) if int(half_tiles) in (4, 8) else HALF_WAVES, _stream()), "mfp_block_fwd_xhat_half")
        else:
            check((lib.mfp_block_fwd_xhat if xhat_stash else lib.mfp_block_fwd)(*args, _stream()), "mfp_block_fwd")
    return x2, (y1, mean1, rstd1, qkv, a, lse, x1, y2, mean2, rstd2, h)


# ---------------------------------------------------------------- d_model 512 (csrc/block_d512.hip)
def ln_dense_d512(x, gamma, beta, W, bias, N: int, relu: bool = False, xhat_stash: bool = False):
    """out = (relu?)(LN(x) W^T + bias) in one launch, d_model 512 (see mfp_ln_dense_d512).  Returns (out, y, mean, rstd);
    ``xhat_stash``: y holds x-hat = (x - mean) rstd instead of LN(x) (mfp_ln_dense_d512_xhat)."""
    lib = load()
    T, D = x.shape
    assert D == 512 and x.dtype == torch.float32 and W.dtype == torch.bfloat16
    dev = x.device
    y = torch.empty((T, D), dtype=torch.bfloat16, device=dev)
    out = torch.empty((T, N), dtype=torch.bfloat16, device=dev)
    mean = torch.empty((T,), dtype=torch.float32, device=dev)
    rstd = torch.empty((T,), dtype=torch.float32, device=dev)
    with _timed("as512_kernel", 2 * T * D * N, T * (D * 4 + D * 2 + N * 2) + N * D * 2):
        check((lib.mfp_ln_dense_d512_xhat if xhat_stash else lib.mfp_ln_dense_d512)(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(W), _ptr(bias), _ptr(y), _ptr(mean), _ptr(rstd),
                                    _ptr(out), T, N, int(relu), LN_EPS, _stream()), "mfp_ln_dense_d512")
    return out, y, mean, rstd


def dense_relumask_d512(A, W, aux):
    """out bf16 [T, N] = (A W^T) * [aux > 0] (see mfp_dense_relumask_d512): A [T, 512], W [N, 512], aux [T, N]."""
    lib = load()
    T, K = A.shape
    N = aux.shape[1]
    assert K == 512 and A.dtype == torch.bfloat16 and aux.dtype == torch.bfloat16 and A.is_contiguous() and aux.is_contiguous()
    out = torch.empty((T, N), dtype=torch.bfloat16, device=A.device)
    with _timed("as512_kernel", 2 * T * K * N, T * (K * 2 + 2 * N * 2) + N * K * 2):
        check(lib.mfp_dense_relumask_d512(_ptr(A), _ptr(W), _ptr(aux), _ptr(out), T, N, _stream()), "mfp_dense_relumask_d512")
    return out


def dense_n512_res(A, W, bias, residual, dropout: Tuple[float, int, int] = (0.0, 0, 0), step_ptr=None, out_bf16=None):
    """out f32 [T, 512] = residual + Dropout(A W^T + bias) (see mfp_dense_n512_res): A bf16 [T, K], W bf16 [512, K]."""
    lib = load()
    T, K = A.shape
    assert A.dtype == torch.bfloat16 and A.is_contiguous() and residual.dtype == torch.float32 and residual.shape == (T, 512)
    out = torch.empty((T, 512), dtype=torch.float32, device=A.device)
    with _timed("os512_kernel", 2 * T * K * 512, T * (K * 2 + 512 * 8 + (512 * 2 if out_bf16 is not None else 0)) + 512 * K * 2):
        check(lib.mfp_dense_n512_res(_ptr(A), _ptr(W), _ptr(bias), _ptr(residual), _ptr(out), _ptr(out_bf16), T, K,
                                     float(dropout[0]), int(dropout[1]), int(dropout[2]),
                                     _ptr(step_ptr) if step_ptr is not None else None, _stream()), "mfp_dense_n512_res")
    return out


def dense_n512(A, W):
    """out bf16 [T, 512] = A W^T (see mfp_dense_n512): A bf16 [T, K], W bf16 [512, K]."""
    lib = load()
    T, K = A.shape
    assert A.dtype == torch.bfloat16 and A.is_contiguous()
    out = torch.empty((T, 512), dtype=torch.bfloat16, device=A.device)
    with _timed("os512_kernel", 2 * T * K * 512, T * (K * 2 + 512 * 2) + 512 * K * 2):
        check(lib.mfp_dense_n512(_ptr(A), _ptr(W), _ptr(out), T, K, _stream()), "mfp_dense_n512")
    return out


def dense_n512_lda(A, W):
    """out bf16 [T, 512] = A W[:, :U]^T for A bf16 [T, U] and W bf16 [512, K], K = U rounded up (W's columns >= U are zero):
    see mfp_dense_n512_lda."""
    lib = load()
    T, U = A.shape
    K = W.shape[1]
    assert A.dtype == torch.bfloat16 and A.is_contiguous() and W.shape[0] == 512 and W.is_contiguous() and K - 128 < U <= K
    out = torch.empty((T, 512), dtype=torch.bfloat16, device=A.device)
    with _timed("os512_kernel", 2 * T * K * 512, T * (U * 2 + 512 * 2) + 512 * K * 2):
        check(lib.mfp_dense_n512_lda(_ptr(A), U, _ptr(W), _ptr(out), T, K, _stream()), "mfp_dense_n512_lda")
    return out


_LNB_WS = {}


def dense_n512_lnb(A, W, xhat, gamma, rstd, dres, dgamma, dbeta, drop=None, jobs: Optional[list] = None):
    """``layernorm_bwd(dense_n512(A, W), ..., xhat=xhat)`` in ONE launch (see mfp_dense_n512_lnb): dy = A W^T never reaches HBM.
    Returns (dx, ddrop) with ``drop`` = (colsum_out [512], p, seed, offset, step_ptr), else dx.  The exchange scratch and its
    flags (zero between launches) are per device and token count, allocated once outside any graph capture's pool."""
    lib = load()
    T, K = A.shape
    assert A.dtype == torch.bfloat16 and A.is_contiguous() and T % 128 == 0 and xhat.dtype == torch.bfloat16 and dres.dtype == torch.bfloat16
    dev = A.device
    # (per stream: two streams stepping models of the same token count must not share the flags / row sums)
    key = (dev.index, T, int(torch.cuda.current_stream(dev).cuda_stream))
    if key not in _LNB_WS:
        assert not torch.cuda.is_current_stream_capturing(), \
            "dense_n512_lnb: first call inside a graph capture (run one warm-up step on the capture stream first: capture_train_step(warmup >= 1))"
        _LNB_WS[key] = (torch.empty((T // 128, 2, 128, 2), dtype=torch.float32, device=dev),
                        torch.zeros((T // 128, 2), dtype=torch.int32, device=dev))
    exch, flags = _LNB_WS[key]
    dx = torch.empty((T, 512), dtype=torch.bfloat16, device=dev)
    ddrop = torch.empty((T, 512), dtype=torch.bfloat16, device=dev) if drop is not None else None
    colsum, p_, seed_, off_, sp_ = drop if drop is not None else (None, 0.0, 0, 0, None)
    P = T // 128
    part = torch.empty((P, 3 * 512), dtype=torch.float32, device=dev)
    with _timed("os512_kernel", 2 * T * K * 512, T * (K * 2 + 512 * 2 * (3 + (1 if drop is not None else 0))) + 512 * K * 2):
        check(lib.mfp_dense_n512_lnb(_ptr(A), _ptr(W), _ptr(xhat), _ptr(gamma), _ptr(rstd), _ptr(dres), _ptr(dx), _ptr(ddrop), _ptr(part),
                                     _ptr(exch), _ptr(flags), T, K, float(p_), int(seed_), int(off_), _ptr(sp_), _stream()),
              "mfp_dense_n512_lnb")
    n = 3 * 512 if drop is not None else 2 * 512
    if jobs is not None:
        jobs.append(dict(part=part, out0=dgamma, out1=dbeta, out2=colsum, split1=512, split2=1024, P=P, N=n, pstride=3 * 512))
    else:
        check(lib.mfp_reduce_partials(part.data_ptr(), _ptr(dgamma), _ptr(dbeta), _ptr(colsum), 512, 1024, P, n, 3 * 512, _stream()),
              "mfp_reduce_partials")
    return (dx, ddrop) if drop is not None else dx


def block_infer(x, gamma, beta, Wqkv, bqkv, Wo, bo, nvalid, gamma2, beta2, W1, b1, W2, b2, B: int, S: int, H: int):
    """A whole DeepSVG block forward in ONE launch with nothing saved for a backward pass (see mfp_block_infer): the
    inference callers' form (``MFP.__call__(training=False)``, ``iterative_decode``, eval.py).  Returns x2."""
    lib = load()
    T, D = x.shape
    dev = x.device
    x1, x2 = (torch.empty((T, D), dtype=torch.float32, device=dev) for _ in range(2))
    stats = torch.empty((4 * T,), dtype=torch.float32, device=dev)
    flops = 2 * T * D * 3 * D + 4 * B * S * S * D + 2 * T * D * D + 2 * 2 * T * D * 2 * D
    with _timed("block_fwd_kernel", flops, T * D * 4 * 5 + 8 * D * D * 2):
        check(lib.mfp_block_infer(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(Wqkv), _ptr(bqkv), _ptr(Wo), _ptr(bo), _ptr(nvalid),
                                  _ptr(gamma2), _ptr(beta2), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(x1), _ptr(stats),
                                  _ptr(x2), B, S, D, H, LN_EPS, _stream()), "mfp_block_infer")
    return x2


def qkv_fused_fwd(x, gamma, beta, W, bias):
    """qkv = LN(x) W^T + bias in one launch (d_model 256, bf16 weights [768][256]).  Returns (qkv, y1, mean, rstd)."""
    lib = load()
    T, D = x.shape
    dev = x.device
    y1 = torch.empty((T, D), dtype=torch.bfloat16, device=dev)
    qkv = torch.empty((T, 3 * D), dtype=torch.bfloat16, device=dev)
    mean = torch.empty((T,), dtype=torch.float32, device=dev)
    rstd = torch.empty((T,), dtype=torch.float32, device=dev)
    with _timed("qkv_fused_kernel", 2 * T * D * 3 * D, T * (D * 4 + D * 2 + 3 * D * 2) + 3 * D * D * 2):
        check(lib.mfp_qkv_fused_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(W), _ptr(bias), _ptr(y1), _ptr(mean), _ptr(rstd),
                                    _ptr(qkv), T, D, LN_EPS, _stream()), "mfp_qkv_fused_fwd")
    return qkv, y1, mean, rstd


def dgrad_qkv(dqkv: torch.Tensor, Wt: torch.Tensor) -> torch.Tensor:
    """dy1 = dqkv Wqkv (bf16 [T,256]) with Wt = the [256][768] transposed shadow of the fused Q|K|V kernel."""
    lib = load()
    T, K = dqkv.shape
    D = K // 3
    dy = torch.empty((T, D), dtype=torch.bfloat16, device=dqkv.device)
    with _timed("dgrad_qkv_kernel", 2 * T * K * D, T * (K + D) * 2 + K * D * 2):
        check(lib.mfp_dgrad_qkv(_ptr(dqkv), _ptr(Wt), _ptr(dy), T, D, _stream()), "mfp_dgrad_qkv")
    return dy


def dgrad_qkv_ln_half(dqkv, Wt, xhat, gamma, rstd, dres, dgamma, dbeta, drop=None, jobs: Optional[list] = None):
    """``layernorm_bwd(dgrad_qkv(dqkv, Wt), ..., xhat=xhat)`` in ONE launch on 64-row tiles (see mfp_dgrad_qkv_ln_half): dy1 never
    reaches HBM.  Returns (dx, ddrop) with ``drop`` = (colsum_out [256], p, seed, offset, step_ptr), else dx."""
    lib = load()
    T, K = dqkv.shape
    D = K // 3
    assert T % 64 == 0 and xhat.dtype == torch.bfloat16 and dres.dtype == torch.bfloat16
    dev = dqkv.device
    dx = torch.empty((T, D), dtype=torch.bfloat16, device=dev)
    ddrop = torch.empty((T, D), dtype=torch.bfloat16, device=dev) if drop is not None else None
    colsum, p_, seed_, off_, sp_ = drop if drop is not None else (None, 0.0, 0, 0, None)
    P = T // 64
    part = torch.empty((P, 3 * D), dtype=torch.float32, device=dev)
    with _timed("dgrad_qkv_kernel", 2 * T * K * D, T * (K * 2 + D * 2 * (3 + (1 if drop is not None else 0))) + K * D * 2):
        check(lib.mfp_dgrad_qkv_ln_half(_ptr(dqkv), _ptr(Wt), _ptr(xhat), _ptr(gamma), _ptr(rstd), _ptr(dres), _ptr(dx), _ptr(ddrop),
                                        _ptr(part), part.numel() * 4, T, D, float(p_), int(seed_), int(off_), _ptr(sp_), _stream()),
              "mfp_dgrad_qkv_ln_half")
    n = 3 * D if drop is not None else 2 * D
    if jobs is not None:
        jobs.append(dict(part=part, out0=dgamma, out1=dbeta, out2=colsum, split1=D, split2=2 * D, P=P, N=n, pstride=3 * D))
    else:
        check(lib.mfp_reduce_partials(part.data_ptr(), _ptr(dgamma), _ptr(dbeta), _ptr(colsum), D, 2 * D, P, n, 3 * D, _stream()),
              "mfp_reduce_partials")
    return (dx, ddrop) if drop is not None else dx


def attn_block_bwd(d_o1, Wot, qkv, a, lse, nvalid, Wqkvt, B: int, S: int, H: int):
    """(dqkv, dy1) of the attention half of a block in ONE launch (see mfp_attn_block_bwd): da = d_o1 Wo stays on chip."""
    lib = load()
    T, D = d_o1.shape
    dqkv = torch.empty((T, 3 * D), dtype=torch.bfloat16, device=d_o1.device)
    dy1 = torch.empty((T, D), dtype=torch.bfloat16, device=d_o1.device)
    flops = 2 * T * D * D + 10 * B * S * S * D + 2 * T * 3 * D * D
    with _timed("attn_block_bwd_kernel", flops, T * (D * 2 * 3 + 3 * D * 2 * 2) + 4 * D * D * 2):
        check(lib.mfp_attn_block_bwd(_ptr(d_o1), _ptr(Wot), _ptr(qkv), _ptr(a), _ptr(lse), _ptr(nvalid), _ptr(Wqkvt),
                                     _ptr(dqkv), _ptr(dy1), B, S, D, H, _stream()), "mfp_attn_block_bwd")
    return dqkv, dy1


def attn_block_bwd_ln(d_o1, Wot, qkv, a, lse, nvalid, Wqkvt, B: int, S: int, H: int, x, gamma, mean, rstd, dres, dgamma, dbeta,
                      drop=None, jobs: Optional[list] = None, xhat=None):
    """:func:`attn_block_bwd` with the backward of LN1 in its epilogue (``mfp_attn_block_bwd_ln``; bf16 residual-gradient
    stream): returns (dqkv, dx, ddrop) -- dy1 never written.  ``drop`` = (colsum_out [D], p, seed, offset, step_ptr) or None
    (block 0: no masked copy, ddrop = None); ``jobs`` as in :func:`mlp_bwd_ln`."""
    lib = load()
    T, D = d_o1.shape
    dev = d_o1.device
    dqkv = torch.empty((T, 3 * D), dtype=torch.bfloat16, device=dev)
    dx = torch.empty((T, D), dtype=torch.bfloat16, device=dev)
    ddrop = torch.empty((T, D), dtype=torch.bfloat16, device=dev) if drop is not None else None
    colsum, p_, seed_, off_, sp_ = drop if drop is not None else (None, 0.0, 0, 0, None)
    P = T // 128
    part = torch.empty((P, 3 * D), dtype=torch.float32, device=dev)
    flops = 2 * T * D * D + 10 * B * S * S * D + 2 * T * 3 * D * D
    nb = T * (D * 2 * 2 + 3 * D * 2 * 2 + D * (4 + 2 + 2 + (2 if drop is not None else 0))) + 4 * D * D * 2
    with _timed("attn_block_bwd_kernel", flops, nb):
        check(lib.mfp_attn_block_bwd_ln(_ptr(d_o1), _ptr(Wot), _ptr(qkv), _ptr(a), _ptr(lse), _ptr(nvalid), _ptr(Wqkvt), _ptr(dqkv),
                                        _ptr(x), _ptr(xhat), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dres), _ptr(dx), _ptr(ddrop), _ptr(part),
                                        part.numel() * 4, B, S, D, H, float(p_), int(seed_), int(off_), _ptr(sp_), _stream()),
              "mfp_attn_block_bwd_ln")
    n = 3 * D if drop is not None else 2 * D
    if jobs is not None:
        jobs.append(dict(part=part, out0=dgamma, out1=dbeta, out2=colsum, split1=D, split2=2 * D, P=P, N=n, pstride=3 * D))
    else:
        check(lib.mfp_reduce_partials(part.data_ptr(), _ptr(dgamma), _ptr(dbeta), _ptr(colsum), D, 2 * D, P, n, 3 * D, _stream()),
              "mfp_reduce_partials")
    return dqkv, dx, ddrop


def encoder_dense2(xs, Ws, biases, codes, h: torch.Tensor) -> torch.Tensor:
    """h += sum_j [codes[j] == 0] (xs[j] Ws[j]^T + biases[j]) for two 512-wide numerical attributes, in place."""
    lib = load()
    T, D = h.shape
    K = xs[0].shape[1]
    with _timed("enc_dense_kernel", 2 * 2 * T * K * D, T * (2 * K * 2 + 2 * D * 4) + 2 * K * D * 2):
        check(lib.mfp_encoder_dense2(_ptr(xs[0]), _ptr(xs[1]), _ptr(Ws[0]), _ptr(Ws[1]), _ptr(biases[0]), _ptr(biases[1]),
                                     _ptr(codes[0]), _ptr(codes[1]), _ptr(h), T, D, K, _stream()), "mfp_encoder_dense2")
    return h


def dgrad_d256(dy: torch.Tensor, Wt: torch.Tensor) -> torch.Tensor:
    """dx = dy W (bf16 [T,256]) for a 256 -> 256 Dense with Wt = its [256][256] transposed shadow."""
    lib = load()
    T, D = dy.shape
    dx = torch.empty((T, D), dtype=torch.bfloat16, device=dy.device)
    with _timed("dgrad_qkv_kernel", 2 * T * D * D, T * 2 * D * 2 + D * D * 2):
        check(lib.mfp_dgrad_d256(_ptr(dy), _ptr(Wt), _ptr(dx), T, D, _stream()), "mfp_dgrad_d256")
    return dx


def dgrad_rows(A: torch.Tensor, Wt: torch.Tensor, K: int, drop: Optional[tuple] = None):
    """C f32 [T,256] = A[:, :K] Wt[:, :K]^T; A bf16 [T][lda], Wt bf16 [256][ldw] zero-padded beyond K (ldw % 128 == 0).
    ``drop`` = (p, seed, offset, step_ptr): also returns the dropout-masked, 1/keep-scaled bf16 copy of C (what
    :func:`dropout_bwd` would produce from it) -> (C, C_drop)."""
    lib = load()
    T = A.shape[0]
    out = torch.empty((T, Wt.shape[0]), dtype=torch.float32, device=A.device)
    out_d = torch.empty((T, Wt.shape[0]), dtype=torch.bfloat16, device=A.device) if drop is not None else None
    p_, seed_, off_, sp_ = drop if drop is not None else (0.0, 0, 0, None)
    with _timed("dgrad_rows_kernel", 2 * T * K * Wt.shape[0], T * (K * 2 + Wt.shape[0] * (4 + (2 if drop else 0))) + Wt.numel() * 2):
        check(lib.mfp_dgrad_rows(_ptr(A), A.stride(0), _ptr(Wt), Wt.stride(0), _ptr(out), T, Wt.shape[0], K,
                                 _ptr(out_d), float(p_), int(seed_), int(off_), _ptr(sp_), _stream()), "mfp_dgrad_rows")
    return out if drop is None else (out, out_d)


def mlp_fused_bwd(d_o2, h, W2t, W1t):
    """dh = (d_o2 W2) * [h > 0], dy2 = dh W1 in one launch (d_model 256, bf16); W2t / W1t are the transposed
    (k-major) shadows.  Returns (dh, dy2)."""
    lib = load()
    T, D = d_o2.shape
    dh = torch.empty((T, 2 * D), dtype=torch.bfloat16, device=d_o2.device)
    dy2 = torch.empty((T, D), dtype=torch.bfloat16, device=d_o2.device)
    with _timed("mlp_bwd_kernel", 2 * 2 * T * D * 2 * D, T * (D * 2 * 2 + 2 * D * 2 * 2) + 2 * D * 2 * D * 2):
        check(lib.mfp_mlp_fused_bwd(_ptr(d_o2), _ptr(h), _ptr(W2t), _ptr(W1t), _ptr(dh), _ptr(dy2), T, D, _stream()),
              "mfp_mlp_fused_bwd")
    return dh, dy2


def mlp_bwd_ln(d_o2, h, W2t, W1t, x, gamma, mean, rstd, dres, dgamma, dbeta, drop, jobs: Optional[list] = None, xhat=None,
               half_tiles: bool = False):
    """:func:`mlp_fused_bwd` with the backward of LN2 in its epilogue (``mfp_mlp_bwd_ln``; bf16 residual-gradient stream):
    returns (dh, dx, ddrop) -- what ``mlp_fused_bwd`` + ``layernorm_bwd(dy2, x, ..., dres, drop=drop)`` return, dy2 never
    written.  ``drop`` = (colsum_out [D], p, seed, offset, step_ptr); ``jobs``: the parameter-gradient partials (one row per
    128-row tile) join the batched reduction at the end of the backward pass, else they are reduced here.  ``xhat``: the bf16
    stash (x - mean) rstd of ``block_fwd(xhat_stash=True)`` -- x and mean are then not read.  ``half_tiles`` (x-hat form):
    two workgroups per 128-row tile (mfp_mlp_bwd_ln_half), one partial row per half tile."""
    lib = load()
    T, D = d_o2.shape
    dev = d_o2.device
    dh = torch.empty((T, 2 * D), dtype=torch.bfloat16, device=dev)
    dx = torch.empty((T, D), dtype=torch.bfloat16, device=dev)
    ddrop = torch.empty((T, D), dtype=torch.bfloat16, device=dev)
    colsum, p_, seed_, off_, sp_ = drop
    assert not half_tiles or xhat is not None
    P = T // 64 if half_tiles else T // 128
    part = torch.empty((P, 3 * D), dtype=torch.float32, device=dev)
    with _timed("mlp_bwd_kernel", 2 * 2 * T * D * 2 * D, T * (D * 2 + 2 * D * 2 * 2 + D * (4 + 2 + 2 + 2)) + 2 * D * 2 * D * 2):
        if half_tiles:
            check(lib.mfp_mlp_bwd_ln_half(_ptr(d_o2), _ptr(h), _ptr(W2t), _ptr(W1t), _ptr(dh), _ptr(xhat), _ptr(gamma), _ptr(rstd),
                                          _ptr(dres), _ptr(dx), _ptr(ddrop), _ptr(part), part.numel() * 4, T, D, float(p_),
                                          int(seed_), int(off_), _ptr(sp_), _stream()), "mfp_mlp_bwd_ln_half")
        else:
          check(lib.mfp_mlp_bwd_ln(_ptr(d_o2), _ptr(h), _ptr(W2t), _ptr(W1t), _ptr(dh), _ptr(x), _ptr(xhat), _ptr(gamma), _ptr(mean),
                                 _ptr(rstd), _ptr(dres), _ptr(dx), _ptr(ddrop), _ptr(part), part.numel() * 4, T, D, float(p_),
                                 int(seed_), int(off_), _ptr(sp_), _stream()), "mfp_mlp_bwd_ln")
    if jobs is not None:
        jobs.append(dict(part=part, out0=dgamma, out1=dbeta, out2=colsum, split1=D, split2=2 * D, P=P, N=3 * D, pstride=3 * D))
    else:
        check(lib.mfp_reduce_partials(part.data_ptr(), _ptr(dgamma), _ptr(dbeta), _ptr(colsum), D, 2 * D, P, 3 * D, 3 * D,
                                      _stream()), "mfp_reduce_partials")
    return dh, dx, ddrop


def layernorm_bwd(dy, x, gamma, mean, rstd, dres: Optional[torch.Tensor], dgamma: torch.Tensor,
                  dbeta: torch.Tensor, dx: Optional[torch.Tensor] = None, drop=None, defer=None,
                  jobs: Optional[list] = None, xhat: Optional[torch.Tensor] = None):
    """``drop`` = (colsum_out [D], p, seed, offset, step_ptr): also return the dropout-masked,
    compute-dtype copy of dx for the consuming Dense backward (fused mfp_dropout_bwd).
    ``defer(fn, *tensors)``: the parameter-gradient reduction (dgamma, dbeta, colsum -- only the
    optimizer needs them) is handed to ``defer`` (StepCtx.on_side) instead of running in line.
    ``jobs``: instead, append the reduction to this list for ONE batched launch at the end of the
    backward pass (:func:`reduce_partials_batch`)."""
    lib = load()
    if xhat is not None:      # the bf16 stash (x - mean) rstd instead of x, mean (mfp_layernorm_bwd_xhat; bf16 residual stream only)
        assert xhat.dtype == torch.bfloat16 and dres is not None and dres.dtype == torch.bfloat16
        x = xhat
    T, D = x.shape
    # the residual gradient stream (dres in, dx out) is f32, or bf16 when the caller carries it in bf16 (res16)
    res16 = (dres is not None and dres.dtype == torch.bfloat16) or (dx is not None and dx.dtype == torch.bfloat16)
    if dx is None:
        dx = torch.empty((T, D), dtype=torch.bfloat16 if res16 else torch.float32, device=x.device)
    assert dres is None or dres.dtype == dx.dtype
    ddrop = torch.empty((T, D), dtype=dy.dtype, device=x.device) if drop is not None else None
    colsum, p_, seed_, off_, sp_ = drop if drop is not None else (None, 0.0, 0, 0, None)
    nbytes = lib.mfp_layernorm_bwd_workspace_bytes(T, D)
    # deferred reduction: the partials must outlive this call -> their own buffer, not the shared one
    own_ws = defer is not None or jobs is not None
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if own_ws else workspace(nbytes, x.device)
    nb = T * D * (_esz(dy) + _esz(x) + (_esz(dx) if dres is not None else 0) + _esz(dx) + (_esz(dy) if drop is not None else 0))
    with _timed("ln_bwd_kernel", 0, nb):
        if xhat is not None:
            check(lib.mfp_layernorm_bwd_xhat(_ptr(dy), _ptr(xhat), _ptr(gamma), _ptr(rstd), _ptr(dres), _ptr(dx),
                                             _ptr(None if own_ws else dgamma), _ptr(None if own_ws else dbeta), ws.data_ptr(), ws.numel(),
                                             T, D, dt_code(dy.dtype), _ptr(ddrop), _ptr(colsum), float(p_), int(seed_), int(off_),
                                             _ptr(sp_), _stream()), "mfp_layernorm_bwd_xhat")
        else:
            check((lib.mfp_layernorm_bwd_res16 if res16 else lib.mfp_layernorm_bwd)(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dres),
                                    _ptr(dx), _ptr(None if own_ws else dgamma),
                                    _ptr(None if own_ws else dbeta), ws.data_ptr(), ws.numel(), T, D,
                                    dt_code(dy.dtype), _ptr(ddrop), _ptr(colsum), float(p_), int(seed_), int(off_),
                                    _ptr(sp_), _stream()), "mfp_layernorm_bwd")
    if jobs is not None:
        P = lib.mfp_layernorm_bwd_partial_rows(T)
        jobs.append(dict(part=ws, out0=dgamma, out1=dbeta, out2=colsum, split1=D, split2=2 * D, P=P,
                         N=3 * D if drop is not None else 2 * D, pstride=3 * D))
    elif defer is not None:
        P = lib.mfp_layernorm_bwd_partial_rows(T)
        n = 3 * D if drop is not None else 2 * D

        def finish():
            check(lib.mfp_reduce_partials(ws.data_ptr(), _ptr(dgamma), _ptr(dbeta), _ptr(colsum), D, 2 * D, P, n,
                                          3 * D, _stream()), "mfp_reduce_partials")
        defer(finish, ws)
    return (dx, ddrop) if drop is not None else dx


def reduce_partials_batch(jobs: list) -> None:
    """One launch for every pending partial reduction (dicts from ``layernorm_bwd(jobs=...)``); the
    list is emptied.  More than MFP_MAX_REDUCE_JOBS entries go out in several launches."""
    lib = load()
    from . import ReduceJob
    while jobs:
        chunk, rest = jobs[:16], jobs[16:]
        arr = (ReduceJob * len(chunk))()
        for i, j in enumerate(chunk):
            arr[i].part, arr[i].out0, arr[i].out1, arr[i].out2 = (j["part"].data_ptr(), _ptr(j["out0"]), _ptr(j["out1"]),
                                                                   _ptr(j["out2"]))
            arr[i].split1, arr[i].split2, arr[i].N, arr[i].pstride, arr[i].P = j["split1"], j["split2"], j["N"], j["pstride"], j["P"]
        with _timed("reduce_partials_batch", 0, sum(j["P"] * j["N"] * 4 for j in chunk)):
            check(lib.mfp_reduce_partials_batch(arr, len(chunk), _stream()), "mfp_reduce_partials_batch")
        jobs[:] = rest


# ------------------------------------------------------------------------------- attention
def attention_fwd(qkv: torch.Tensor, nvalid: torch.Tensor, B: int, S: int, H: int):
    lib = load()
    D = qkv.shape[1] // 3
    out = torch.empty((B * S, D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=qkv.device)
    with _timed("attn_fwd", 4 * B * S * S * D, B * S * 4 * D * _esz(qkv)):
        check(lib.mfp_attention_fwd(_ptr(qkv), _ptr(nvalid), _ptr(out), _ptr(lse), B, S, H, D // H,
                                    dt_code(qkv.dtype), _stream()), "mfp_attention_fwd")
    return out, lse


def attention_bwd(qkv, nvalid, out, dout, lse, B: int, S: int, H: int) -> torch.Tensor:
    lib = load()
    D = qkv.shape[1] // 3
    dqkv = torch.empty_like(qkv)
    with _timed("attn_bwd", 8 * B * S * S * D, B * S * 8 * D * _esz(qkv)):
        check(lib.mfp_attention_bwd(_ptr(qkv), _ptr(nvalid), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv),
                                    B, S, H, D // H, dt_code(qkv.dtype), _stream()), "mfp_attention_bwd")
    return dqkv


# ------------------------------------------------------------------------------- embedding
def embed_pool_fwd(idx: torch.Tensor, rowoff: torch.Tensor, tables: torch.Tensor) -> torch.Tensor:
    lib = load()
    T, NCOL = idx.shape
    ROWS, D = tables.shape
    out = torch.empty((T, D), dtype=torch.float32, device=idx.device)
    with _timed("embed_fwd_kernel", 0, T * (NCOL * 4 + D * 4)):
        check(lib.mfp_embed_pool_fwd(_ptr(idx), _ptr(rowoff), _ptr(tables), _ptr(out), T, NCOL, ROWS, D,
                                     _stream()), "mfp_embed_pool_fwd")
    return out


def embed_pool_bwd(idx, rowoff, dout: torch.Tensor, dtables: torch.Tensor) -> torch.Tensor:
    lib = load()
    T, NCOL = idx.shape
    ROWS, D = dtables.shape
    nbytes = lib.mfp_embed_pool_bwd_workspace_bytes(T, NCOL, ROWS, D)
    ws = workspace(nbytes, idx.device)
    with _timed("embed_bwd_kernel", 0, T * (NCOL * 4 + D * 4)):
        check(lib.mfp_embed_pool_bwd(_ptr(idx), _ptr(rowoff), _ptr(dout), _ptr(dtables), ws.data_ptr(),
                                     ws.numel(), T, NCOL, ROWS, D, _stream()), "mfp_embed_pool_bwd")
    return dtables


def embed_onehot(idx: torch.Tensor, rowoff: torch.Tensor, rows_pad: int) -> torch.Tensor:
    lib = load()
    T, ncol = idx.shape
    P = torch.empty((T, rows_pad), dtype=torch.bfloat16, device=idx.device)
    with _timed("embed_fwd_kernel", 0, T * rows_pad * 2):
        check(lib.mfp_embed_onehot(_ptr(idx), _ptr(rowoff), _ptr(P), T, ncol, rows_pad, _stream()), "mfp_embed_onehot")
    return P


def row_flags(x: torch.Tensor, rowcode: torch.Tensor, special_idx: Optional[torch.Tensor] = None,
              idx_stride: int = 1):
    """x f32 [T,K] -> rowcode u8 [T]; optionally special_idx[t*stride] = rowcode-1."""
    lib = load()
    T, K = x.shape
    with _timed("row_flags_kernel", 0, T * K * 4):
        check(lib.mfp_row_flags(_ptr(x), _ptr(rowcode), _ptr(special_idx), idx_stride, T, K, _stream()),
              "mfp_row_flags")
    return rowcode


# ---------------------------------------------------------------------------------- losses
def loss_fwd_bwd(logits: torch.Tensor, keys: Sequence[dict], nvalid: torch.Tensor, B: int, S: int,
                 dl_dtype: Optional[torch.dtype], sums: Optional[torch.Tensor] = None,
                 dlogits: Optional[torch.Tensor] = None, pred_row: Optional[torch.Tensor] = None,
                 true_row: Optional[torch.Tensor] = None, prezeroed: bool = False):
    """keys: dicts with col_off, n_feat, n_class, is_numerical, target, mask, cond_idx,
    cond_stride, cond_bits.  Returns (sums [nkeys,3], dlogits or None).  ``pred_row`` /
    ``true_row`` (int32 [B*S] permutations from :func:`sort_positions`) select the RICO
    position-sorted loss (reference metrics.py:180-211).  ``prezeroed``: ``sums`` was zeroed by
    :func:`step_prologue`; accumulate into it without a zeroing launch."""
    lib = load()
    ld = logits.shape[1]
    arr = (LossKey * len(keys))()
    for i, k in enumerate(keys):
        arr[i].col_off, arr[i].n_feat, arr[i].n_class = k["col_off"], k["n_feat"], k["n_class"]
        arr[i].is_numerical = int(k["is_numerical"])
        arr[i].target, arr[i].mask = _ptr(k["target"]), _ptr(k["mask"])
        arr[i].cond_idx = _ptr(k.get("cond_idx"))
        arr[i].cond_stride = k.get("cond_stride", 1)
        arr[i].cond_bits = k.get("cond_bits", 0xFFFFFFFF)
    if sums is None:
        sums = torch.empty((len(keys), 3), dtype=torch.float32, device=logits.device)
    if dl_dtype is not None and dlogits is None:
        dlogits = torch.zeros(logits.shape, dtype=dl_dtype, device=logits.device)
    code = dt_code(dlogits.dtype) if dlogits is not None else MFP_F32
    nb = logits.numel() * (4 + (_esz(dlogits) if dlogits is not None else 0))
    with _timed("loss_kernels(ce+mse)", 0, nb):
        if prezeroed:
            check(lib.mfp_loss_fwd_bwd_acc(_ptr(logits), _ptr(dlogits), ld, arr, len(keys), _ptr(nvalid), _ptr(sums),
                                           B, S, code, _ptr(pred_row), _ptr(true_row), _stream()), "mfp_loss_fwd_bwd_acc")
        elif pred_row is None and true_row is None:
            check(lib.mfp_loss_fwd_bwd(_ptr(logits), _ptr(dlogits), ld, arr, len(keys), _ptr(nvalid), _ptr(sums),
                                       B, S, code, _stream()), "mfp_loss_fwd_bwd")
        else:
            for m in (pred_row, true_row):
                assert m is None or (m.dtype == torch.int32 and m.numel() == B * S and m.is_contiguous())
            check(lib.mfp_loss_fwd_bwd_sorted(_ptr(logits), _ptr(dlogits), ld, arr, len(keys), _ptr(nvalid),
                                              _ptr(sums), B, S, code, _ptr(pred_row), _ptr(true_row), _stream()),
                  "mfp_loss_fwd_bwd_sorted")
    return sums, dlogits


def _loss_key_array(keys: Sequence[dict]):
    arr = (LossKey * len(keys))()
    for i, k in enumerate(keys):
        arr[i].col_off, arr[i].n_feat, arr[i].n_class = k["col_off"], k["n_feat"], k["n_class"]
        arr[i].is_numerical = int(k["is_numerical"])
        arr[i].target, arr[i].mask = _ptr(k["target"]), _ptr(k["mask"])
        arr[i].cond_idx = _ptr(k.get("cond_idx"))
        arr[i].cond_stride = k.get("cond_stride", 1)
        arr[i].cond_bits = k.get("cond_bits", 0xFFFFFFFF)
    return arr


def heads_loss_fused_ok(keys: Sequence[dict], U: int, D: int, T: int = 0, want_logits: bool = True) -> bool:
    """Shapes mfp_heads_loss_fwd_bwd takes: d_model 256, 8-aligned heads, categorical items of <= 64 classes on 8-column
    boundaries, numerical widths % 8 == 0, at most 16 (key, feature) items and 40 column chunks, and a token count whose
    logits / d(logits) rows stay inside the kernel's 32-bit byte offsets (beyond: the four-launch path)."""
    if D != 256 or U % 8 != 0 or U > 1536 or len(keys) > 16:
        return False
    if T > (1 << 20) or T * U * (4 if want_logits else 2) >= 0xFFFFFFF0:
        return False
    items = chunks = 0
    for k in keys:
        if k["col_off"] % 8 != 0:
            return False
        if k["is_numerical"]:
            if k["n_feat"] != 1 or k["n_class"] % 8 != 0:
                return False
            chunks += (k["n_class"] + 63) // 64
        else:
            if k["n_class"] > 64 or (k["n_feat"] > 1 and k["n_class"] % 8 != 0):
                return False
            items += k["n_feat"]
            chunks += k["n_feat"]
    return items <= 16 and chunks <= 40


# heads + loss launch on 64-row tiles when the 128-row tiles do not fill the chip (c4: 128 documents per GPU); unset = by grid
# size, "0" / "1" = A/B
HEADS_HALF = os.environ.get("MFP_HEADS_HALF", "")


def heads_half_on(T: int, device=None) -> bool:
    if HEADS_HALF != "":
        return HEADS_HALF == "1"
    return 2 * ((T + 127) // 128) <= cu_count(device)


def heads_loss_fused(x_c: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, keys: Sequence[dict], nvalid: torch.Tensor,
                     B: int, S: int, dlogits: Optional[torch.Tensor] = None, want_logits: bool = True,
                     drop: Optional[tuple] = None, dx_dtype: torch.dtype = torch.float32, half_tiles: Optional[bool] = None):
    """Heads forward + LossLayer + heads input gradient in ONE launch (see mfp_heads_loss_fwd_bwd).  x_c bf16 [B*S,256],
    W bf16 [U,256], bias f32 [U].  Returns (part [P,48] per-workgroup partial sums, dlogits bf16 [T,U], logits f32 [T,U] or
    None, dx f32 [T,256], dx_drop bf16 [T,256] or None); ``drop`` = (p, seed, offset, step_ptr) as in :func:`dgrad_rows`.
    Reduce the partials with :func:`reduce_partials` (N = 3 * len(keys), pstride 48)."""
    lib = load()
    T, D = x_c.shape
    U = W.shape[0]
    dev = x_c.device
    arr = _loss_key_array(keys)
    half = heads_half_on(T, dev) if half_tiles is None else half_tiles
    P = lib.mfp_heads_loss_partials_half(T) if half else lib.mfp_heads_loss_partials(T)
    part = torch.empty((P, 48), dtype=torch.float32, device=dev)
    if dlogits is None:
        dlogits = torch.empty((T, U), dtype=torch.bfloat16, device=dev)
    logits = torch.empty((T, U), dtype=torch.float32, device=dev) if want_logits else None
    dx = torch.empty((T, D), dtype=dx_dtype, device=dev)      # f32, or bf16 for a step that carries residual gradients in bf16
    dxd = torch.empty((T, D), dtype=torch.bfloat16, device=dev) if drop is not None else None
    p_, seed_, off_, sp_ = drop if drop is not None else (0.0, 0, 0, None)
    nb = T * (D * 2 + U * 2 + D * _esz(dx) + (D * 2 if drop is not None else 0) + (U * 4 if want_logits else 0)) + U * D * 2
    with _timed("heads_loss_kernel", 2 * 2 * T * U * D, nb):
        fn = lib.mfp_heads_loss_fwd_bwd_half if half else lib.mfp_heads_loss_fwd_bwd
        check(fn(_ptr(x_c), _ptr(W), _ptr(bias), U, arr, len(keys), _ptr(nvalid), _ptr(part),
                 _ptr(dlogits), _ptr(logits), _ptr(dx) if dx_dtype == torch.float32 else None,
                 _ptr(dx) if dx_dtype == torch.bfloat16 else None, _ptr(dxd), B, S, D, float(p_), int(seed_),
                 int(off_), _ptr(sp_), _stream()), "mfp_heads_loss_fwd_bwd")
    return part, dlogits, logits, dx, dxd


def reduce_partials(part: torch.Tensor, out: torch.Tensor, N: int) -> None:
    """out[:N] = column sums of part[:, :N] (one launch; fixed summation order)."""
    lib = load()
    with _timed("reduce_partials_batch", 0, part.numel() * 4):
        check(lib.mfp_reduce_partials(_ptr(part), _ptr(out), None, None, N, N, part.shape[0], N, part.shape[1], _stream()),
              "mfp_reduce_partials")


def sort_positions(nvalid: torch.Tensor, flag: torch.Tensor, B: int, S: int, labels: Sequence[torch.Tensor] = None,
                   logits: torch.Tensor = None, heads: Sequence = None, out: torch.Tensor = None) -> torch.Tensor:
    """Row map of ``sort_inputs`` (reference tensor_utils.py:14-44) for the documents whose
    ``flag`` (uint8 [B]) is set, identity elsewhere.  ``labels``: five int32 tensors [B*S, n]
    (type, left, top, width, height; feature 0 is the sort key) -- or ``logits`` f32 [B*S, ld] with
    ``heads`` = five (col_off, n_class) pairs (from_logits=True: first-index argmax)."""
    lib = load()
    assert flag.dtype == torch.uint8 and flag.numel() == B and nvalid.dtype == torch.int32
    if out is None:
        out = torch.empty(B * S, dtype=torch.int32, device=nvalid.device)
    if logits is not None:
        assert logits.dtype == torch.float32 and logits.is_contiguous() and len(heads) == 5
        co = (ctypes.c_int32 * 5)(*[int(h[0]) for h in heads])
        nc = (ctypes.c_int32 * 5)(*[int(h[1]) for h in heads])
        check(lib.mfp_sort_positions(None, None, _ptr(logits), logits.shape[1], co, nc, _ptr(nvalid), _ptr(flag),
                                     _ptr(out), B, S, _stream()), "mfp_sort_positions")
    else:
        assert len(labels) == 5
        for t in labels:
            assert t.dtype == torch.int32 and t.is_contiguous() and t.numel() % (B * S) == 0
        ptrs = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in labels])
        strides = (ctypes.c_int32 * 5)(*[t.numel() // (B * S) for t in labels])
        check(lib.mfp_sort_positions(ptrs, strides, None, 0, None, None, _ptr(nvalid), _ptr(flag), _ptr(out),
                                     B, S, _stream()), "mfp_sort_positions")
    return out


# ------------------------------------------------------------------------------- optimizer
class AdamChunks:
    """Static chunk table of a flat parameter buffer (see mfp_adam_chunk_table)."""

    def __init__(self, seg_off: Sequence[int], device):
        lib = load()
        nseg = len(seg_off) - 1
        off = (ctypes.c_int32 * (nseg + 1))(*seg_off)
        n = lib.mfp_adam_num_chunks(off, nseg)
        cs, cb, cl = (ctypes.c_int32 * n)(), (ctypes.c_int64 * n)(), (ctypes.c_int32 * n)()
        sf = (ctypes.c_int32 * (nseg + 1))()
        check(lib.mfp_adam_chunk_table(off, nseg, cs, cb, cl, sf), "mfp_adam_chunk_table")
        self.nseg, self.nchunks = nseg, n
        self.seg_first = torch.tensor(list(sf), dtype=torch.int32, device=device)
        self.partial = torch.zeros((n, 2), dtype=torch.float32, device=device)
        self.chunk_seg = torch.tensor(list(cs), dtype=torch.int32, device=device)
        self.chunk_beg = torch.tensor(list(cb), dtype=torch.int64, device=device)
        self.chunk_len = torch.tensor(list(cl), dtype=torch.int32, device=device)


def adam_keras(w, g, m, v, shadow: Optional[torch.Tensor], chunks: AdamChunks, seg_l2: torch.Tensor,
               stats: torch.Tensor, step_t: torch.Tensor, lr: float, beta1: float = 0.9,
               beta2: float = 0.999, eps: float = 1e-7, clipnorm: float = 1.0, grad_scale: float = 1.0):
    lib = load()
    with _timed("adam_kernels", 0, w.numel() * (8 + 16 + 12 + (2 if shadow is not None else 0))):
      check(lib.mfp_adam_keras(_ptr(w), _ptr(g), _ptr(m), _ptr(v), _ptr(shadow), _ptr(chunks.chunk_seg),
                             _ptr(chunks.chunk_beg), _ptr(chunks.chunk_len), _ptr(chunks.seg_first), chunks.nchunks,
                             _ptr(seg_l2), _ptr(chunks.partial), _ptr(stats), chunks.nseg, _ptr(step_t), lr, beta1, beta2, eps,
                             clipnorm if clipnorm is not None else 0.0, grad_scale, _stream()),
          "mfp_adam_keras")


def cast_bf16(src: torch.Tensor, dst: torch.Tensor):
    lib = load()
    with _timed("cast_kernel", 0, src.numel() * 6):
        check(lib.mfp_cast_f32_bf16(_ptr(src), _ptr(dst), src.numel(), _stream()), "mfp_cast_f32_bf16")
    return dst


class TransposeTable:
    """Device table of the [rows][cols] matrices (flat offset, rows, cols[, out offset, out row
    stride]) whose transposed bf16 copy mfp_transpose_cast_bf16 maintains."""

    def __init__(self, segs, device):
        segs = [tuple(s) + ((s[0], s[1]) if len(s) == 3 else ()) for s in segs]
        self.nseg = len(segs)
        self.off = torch.tensor([s[0] for s in segs], dtype=torch.int64, device=device)
        self.rows = torch.tensor([s[1] for s in segs], dtype=torch.int32, device=device)
        self.cols = torch.tensor([s[2] for s in segs], dtype=torch.int32, device=device)
        self.ooff = torch.tensor([s[3] for s in segs], dtype=torch.int64, device=device)
        self.old = torch.tensor([s[4] for s in segs], dtype=torch.int32, device=device)
        self.max_tiles = max(((s[1] + 31) // 32) * ((s[2] + 31) // 32) for s in segs)


def transpose_cast_bf16(w: torch.Tensor, out: torch.Tensor, table: TransposeTable):
    lib = load()
    with _timed("cast_kernel", 0, 0):
        check(lib.mfp_transpose_cast_bf16(_ptr(w), _ptr(out), _ptr(table.off), _ptr(table.rows), _ptr(table.cols),
                                          _ptr(table.ooff), _ptr(table.old), table.nseg, table.max_tiles, _stream()),
              "mfp_transpose_cast_bf16")
    return out


def dropout_bwd(dx: torch.Tensor, out_dtype: torch.dtype, colsum: torch.Tensor, p: float, seed: int,
                offset: int, step_ptr: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = load()
    M, N = dx.shape
    dy = torch.empty((M, N), dtype=out_dtype, device=dx.device)
    ws = workspace(lib.mfp_colsum_workspace_bytes(M, N), dx.device)
    if dx.dtype == torch.bfloat16:      # the residual gradient travels in bf16 (functions.RES_GRAD_BF16)
        assert out_dtype == torch.bfloat16
        with _timed("dropout_bwd_kernel", 0, M * N * 4):
            check(lib.mfp_dropout_bwd_res16(_ptr(dx), _ptr(dy), _ptr(colsum), ws.data_ptr(), ws.numel(), M, N, float(p),
                                            int(seed), int(offset), _ptr(step_ptr), _stream()), "mfp_dropout_bwd_res16")
        return dy
    with _timed("dropout_bwd_kernel", 0, M * N * (4 + _esz(dy))):
        check(lib.mfp_dropout_bwd(_ptr(dx), _ptr(dy), _ptr(colsum), ws.data_ptr(), ws.numel(), M, N, float(p),
                                  int(seed), int(offset), _ptr(step_ptr), dt_code(out_dtype), _stream()),
              "mfp_dropout_bwd")
    return dy


def colsum(X: torch.Tensor, out: torch.Tensor, M: int, N: int, ld: Optional[int] = None) -> torch.Tensor:
    lib = load()
    ws = workspace(lib.mfp_colsum_workspace_bytes(M, N), X.device)
    check(lib.mfp_colsum(_ptr(X), _ptr(out), ws.data_ptr(), ws.numel(), M, N, ld or N, dt_code(X.dtype),
                         _stream()), "mfp_colsum")
    return out


def sample_tasks(probs: Sequence[float], B: int, seed: int, offset: int, step_ptr: Optional[torch.Tensor],
                 device) -> torch.Tensor:
    """int32 [B] ~ Categorical(probs) in one launch (torch.multinomial is ~8 tiny kernels)."""
    lib = load()
    tasks = torch.empty((B,), dtype=torch.int32, device=device)
    arr = (ctypes.c_float * len(probs))(*[float(x) for x in probs])
    check(lib.mfp_sample_tasks(arr, len(probs), _ptr(tasks), B, int(seed), int(offset), _ptr(step_ptr), _stream()),
          "mfp_sample_tasks")
    return tasks


def step_prologue(probs: Sequence[float], length: torch.Tensor, seed: int, offset: int, step_ptr: Optional[torch.Tensor],
                  zero: Optional[torch.Tensor]):
    """(tasks int32 [B] ~ Categorical(probs), nvalid int32 [B] = length + 1) and ``zero[:] = 0`` in ONE launch
    (:func:`sample_tasks` + the ``length + 1`` element-wise op + the loss kernels' zeroing launch)."""
    lib = load()
    assert length.dtype == torch.int32 and length.is_contiguous()
    B = length.numel()
    tasks = torch.empty((B,), dtype=torch.int32, device=length.device)
    nvalid = torch.empty((B,), dtype=torch.int32, device=length.device)
    arr = (ctypes.c_float * len(probs))(*[float(x) for x in probs])
    check(lib.mfp_step_prologue(arr, len(probs), _ptr(tasks), _ptr(length), _ptr(nvalid), B, int(seed), int(offset),
                                _ptr(step_ptr), _ptr(zero), zero.numel() if zero is not None else 0, _stream()),
          "mfp_step_prologue")
    return tasks, nvalid


def mask_tokens(cols: Sequence[dict], idx_all: torch.Tensor, nvalid: torch.Tensor, tasks: torch.Tensor,
                B: int, S: int, seed: int, offset: int, step_ptr: Optional[torch.Tensor], x_dtype: torch.dtype):
    """Fused preprocess_for_train (see mfp_mask_tokens).  cols: dicts with is_numerical, n_feat,
    input_dim, group, src, cond_idx, cond_stride, cond_bits, idx_col, x_out, rowcode, mask_out."""
    lib = load()
    arr = (MaskCol * len(cols))()
    nbytes = 0
    for i, c in enumerate(cols):
        a = arr[i]
        a.is_numerical, a.n_feat, a.input_dim, a.group = int(c["is_numerical"]), c["n_feat"], c.get("input_dim", 0), c["group"]
        a.src, a.cond_idx = _ptr(c["src"]), _ptr(c.get("cond_idx"))
        a.cond_stride, a.cond_bits = c.get("cond_stride", 1), c.get("cond_bits", 0xFFFFFFFF)
        a.idx_col = c["idx_col"]
        a.x_out, a.rowcode, a.mask_out = _ptr(c.get("x_out")), _ptr(c.get("rowcode")), _ptr(c["mask_out"])
        if c["is_numerical"]:
            nbytes += B * S * c["n_feat"] * (4 + c["x_out"].element_size())
    with _timed("mask_kernel", 0, nbytes + idx_all.numel() * 8):
        check(lib.mfp_mask_tokens(arr, len(cols), _ptr(idx_all), idx_all.shape[1], _ptr(nvalid), _ptr(tasks), B, S,
                                  int(seed), int(offset), _ptr(step_ptr), dt_code(x_dtype), _stream()),
              "mfp_mask_tokens")


def tr_probe(byte_addr: torch.Tensor) -> torch.Tensor:
    lib = load()
    out = torch.empty((64, 4), dtype=torch.int16, device=byte_addr.device)
    check(lib.mfp_debug_tr_probe(_ptr(byte_addr), _ptr(out), _stream()), "mfp_debug_tr_probe")
    return out
