"""Single-node data parallelism: one process per GPU, RCCL (torch.distributed "nccl") over xGMI.

The reference has no distributed code (train.py:25 is a commented-out MirroredStrategy); this
is new.  Documents are independent, so the minibatch is sharded on the batch axis with no
data-path collective; the only exchange is ONE sum all-reduce of the flat f32 gradient buffer
per step.  Semantics match the single-GPU step on the global batch:

* each rank's LossLayer uses mean over its B/N documents -> averaging gradients over ranks
  equals the global mean (equal shards);
* the L2 regulariser gradient and the per-variable clipnorm are applied AFTER averaging
  (inside the fused Adam kernel, via ``grad_scale = 1/N``);
* metric numerators/denominators and loss sums are all-reduced for reporting.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> int:
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun); returns world size."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 1
    # the plan switches are read when the step is captured: a typo should fail here, not minutes into a run
    bucket_cut_blocks(2)
    BucketReducer()
    graph_mode()
    if not dist.is_initialized():
        # MFP_DIST_BACKEND=gloo lets several ranks share ONE GPU (tests of the multi-rank step logic);
        # production is "nccl" (= RCCL on ROCm), one rank per GPU over xGMI.
        backend = backend or os.environ.get("MFP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
    reserve_cus_from_env()
    return dist.get_world_size()


def reserve_cus_from_env() -> int:
    """``MFP_DP_RESERVE_CUS=n``: with more than one rank, the persistent one-workgroup-per-CU launches (grouped weight
    gradients, weight-stationary products, single-pass attention backward) size their grids for #CUs - n, leaving n
    CUs to RCCL's workgroups while a bucket's all-reduce runs under the next segment of the backward pass (they cannot
    share a CU with a 160 KB / 8-wave workgroup).  Default 0: an A/B switch for the driver's scaling run.  Returns n."""
    n = int(os.environ.get("MFP_DP_RESERVE_CUS", "0") or 0)
    if n and torch.cuda.is_available():
        from mfp.hip import ops
        ops.set_reserved_cus(n)
    return n


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def active() -> bool:
    """The step runs its data-parallel form: more than one rank -- or ONE rank of an initialised process group under
    MFP_DP_FORCE=1 (tests: the bucketed step, its all-reduces and their hipGraph capture on the one GPU a test box has)."""
    if world_size() > 1:
        return True
    return os.environ.get("MFP_DP_FORCE", "") == "1" and dist.is_available() and dist.is_initialized()


def graph_mode() -> str:
    """``MFP_DP_GRAPH``: how the N > 1 step is replayed.  "one" = the whole step, bucket all-reduces included, is ONE hipGraph
    (the collectives are captured on the communication stream as branches that run beside the next segment of the backward
    pass: one graph launch per step, no host between a segment and its all-reduce); "segments" = one graph per backward
    segment + one for Adam with the all-reduces launched eagerly between them (L + 1 graph launches per step; rounds 2-5);
    "auto" (default) = "one" on the nccl (= RCCL) backend, falling back to "segments" when the capture is refused, and
    "segments" on gloo (host-side collectives cannot be captured)."""
    m = os.environ.get("MFP_DP_GRAPH", "auto")
    if m not in ("auto", "one", "segments"):
        raise ValueError("MFP_DP_GRAPH=%r (auto | one | segments)" % m)
    return m


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def rank_seed(seed: int) -> int:
    """Seed of this rank's counter-based random streams (task draw, masking, dropout): every rank
    must draw independently for its documents, otherwise only B/N of the global batch's draws are
    independent.  Rank 0 keeps ``seed``; the parameter-initialisation seed is NOT rank-dependent."""
    return (int(seed) + 0x9E3779B1 * rank()) & 0x7FFFFFFFFFFFFFFF


def allreduce_gradients(flat_grad: torch.Tensor, async_op: bool = False):
    """Sum the flat gradient over ranks in place.  The 1/N factor is folded into the optimizer
    (``AdamKeras.step(grad_scale=1/N)``) so no extra pass over the buffer is needed."""
    if world_size() == 1:
        return None
    return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=async_op)


def bucket_cut_blocks(num_blocks: int, mode: Optional[str] = None) -> List[int]:
    """Blocks at whose INPUT the data-parallel step cuts its backward pass, in the order the backward pass
    reaches them.  ``"blocks"`` (default): every block > 0 -> buckets {heads + block L-1}, {block L-2}, ...,
    {block 0 + encoder}, each all-reduced while the next segment of the backward pass runs (only the last one and
    Adam are exposed); ``"halves"``: one cut at block L/2 (rounds 1-2); ``"none"``: one all-reduce after the backward
    pass.  MFP_DP_BUCKETS overrides."""
    mode = mode or os.environ.get("MFP_DP_BUCKETS", "blocks")
    if mode == "none" or num_blocks < 2:
        return []
    if mode == "halves":
        return [num_blocks // 2]
    if mode != "blocks":
        raise ValueError("MFP_DP_BUCKETS=%r (blocks | halves | none)" % mode)
    return list(range(num_blocks - 1, 0, -1))


def bucket_slices(offsets: List[int], numel: int) -> List[slice]:
    """Flat-gradient slices of the buckets for descending cut offsets: [off_0, numel), [off_1, off_0), ..., [0, off_last)."""
    out, hi = [], numel
    for off in offsets:
        assert 0 <= off <= hi
        out.append(slice(off, hi))
        hi = off
    out.append(slice(0, hi))
    return out


def describe_plan(layout, graphed: bool = True) -> dict:
    """The gradient all-reduce plan of the step as it will run (bench.py prints it on the N > 1 line): buckets in
    backward order with their bytes on the wire, the carrier dtype, and which part is exposed."""
    mode = os.environ.get("MFP_DP_BUCKETS", "blocks")
    carrier = "bf16" if os.environ.get("MFP_DP_GRAD_DTYPE", "f32") in ("bf16", "bfloat16") else "f32"
    esize = 2 if carrier == "bf16" else 4
    cuts = bucket_cut_blocks(layout.L, mode) if graphed else []
    slices = bucket_slices([layout.block_offset(i) for i in cuts], layout.numel)
    names = []
    for k, sl in enumerate(slices):
        if not cuts:
            names.append("all parameters")
        elif k == 0:
            names.append("heads + block %d" % cuts[0] + ("..%d" % (layout.L - 1) if cuts[0] < layout.L - 1 else ""))
        elif k == len(slices) - 1:
            names.append("encoder + block 0" + ("..%d" % (cuts[-1] - 1) if cuts[-1] > 1 else ""))
        else:
            names.append("block %d" % cuts[k] + ("..%d" % (cuts[k - 1] - 1) if cuts[k - 1] - 1 > cuts[k] else ""))
    return {"buckets": mode if graphed else "none (eager step: one all-reduce after the backward pass)",
            "carrier": carrier,
            "plan": [{"bucket": n, "params": sl.stop - sl.start, "bytes": (sl.stop - sl.start) * esize}
                     for n, sl in zip(names, slices)],
            "bytes_per_step": layout.numel * esize,
            "reserved_cus": int(os.environ.get("MFP_DP_RESERVE_CUS", "0") or 0),
            "exposed": "the last bucket's all-reduce + Adam" if cuts else "the whole all-reduce + Adam"}


class BucketReducer:
    """Sum all-reduce of one gradient bucket, optionally carried as bf16 (MFP_DP_GRAD_DTYPE=bf16: half the xGMI bytes;
    every rank receives the same bf16 sums, so the replicas stay bit-identical; default f32).  ``launch`` is
    asynchronous (the collective runs on the communication stream under whatever the caller enqueues next);
    ``finish`` waits and, for bf16, writes the sums back into the f32 bucket."""

    def __init__(self, grad_dtype: Optional[str] = None):
        name = grad_dtype or os.environ.get("MFP_DP_GRAD_DTYPE", "f32")
        if name not in ("f32", "fp32", "float32", "bf16", "bfloat16"):
            raise ValueError("MFP_DP_GRAD_DTYPE=%r (f32 | bf16)" % name)
        self.bf16 = name in ("bf16", "bfloat16")
        self.pending = []

    def launch(self, bucket: torch.Tensor):
        if not active() or bucket.numel() == 0:
            return
        if self.bf16:
            tmp = bucket.to(torch.bfloat16)
            self.pending.append((dist.all_reduce(tmp, op=dist.ReduceOp.SUM, async_op=True), tmp, bucket))
        else:
            self.pending.append((dist.all_reduce(bucket, op=dist.ReduceOp.SUM, async_op=True), None, bucket))

    def finish(self):
        for work, tmp, bucket in self.pending:
            work.wait()
            if tmp is not None:
                bucket.copy_(tmp)
        self.pending = []


def allreduce_sums(sums: torch.Tensor) -> torch.Tensor:
    """Loss / score sums for metrics: loss column is a per-rank batch mean -> average it;
    score numerators and denominators are plain sums."""
    n = world_size()
    if n == 1:
        return sums
    out = sums.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    out[:, 0] /= n
    return out


def broadcast_parameters(flat_w: torch.Tensor, src: int = 0):
    if world_size() > 1:
        dist.broadcast(flat_w, src=src)


def shard_bounds(B: int, r: int, n: int):
    """[lo, hi) of rank r's documents when B documents are dealt to n ranks as evenly as possible
    (the first B % n ranks hold one more; a rank's shard may be empty when B < n)."""
    q, rem = divmod(B, n)
    lo = r * q + min(r, rem)
    return lo, lo + q + (1 if r < rem else 0)


def shard_batch(batch: dict, r: Optional[int] = None, n: Optional[int] = None, even: bool = True) -> dict:
    """Rank r's slice of a global batch (axis 0).  ``even=True`` (training: the averaged gradient
    equals the global-batch gradient only for equal shards) insists on B % n == 0; ``even=False``
    (evaluation: sums are all-reduced with the document counts) deals a ragged batch out unevenly."""
    r = rank() if r is None else r
    n = world_size() if n is None else n
    if n == 1:
        return batch
    out = {}
    for k, v in batch.items():
        B = len(v)
        if even:
            assert B % n == 0, ("global batch of %d documents is not divisible by the %d data-parallel ranks "
                                "(choose --batch_size as a multiple of the world size)" % (B, n))
        lo, hi = shard_bounds(B, r, n)
        out[k] = v[lo:hi]
    return out


def allreduce_eval_sums(sums: Optional[torch.Tensor], b_local: int, nkeys: int, device) -> torch.Tensor:
    """Evaluation sums of one global batch from the per-rank shards: ``sums [nkeys][3]`` holds the
    shard's (mean-over-documents loss, score numerator, score denominator) or is None for an empty
    shard.  Returns the same triple for the GLOBAL batch (loss = mean over all its documents)."""
    pack = torch.zeros(nkeys * 3 + 1, dtype=torch.float64, device=device)
    if sums is not None and b_local > 0:
        s = sums.to(torch.float64).clone()
        s[:, 0] *= b_local                      # mean over the shard -> sum over its documents
        pack[:-1] = s.reshape(-1)
        pack[-1] = b_local
    if world_size() > 1:
        dist.all_reduce(pack, op=dist.ReduceOp.SUM)
    out = pack[:-1].reshape(nkeys, 3).clone()
    out[:, 0] /= pack[-1].clamp(min=1.0)
    return out.to(torch.float32)
