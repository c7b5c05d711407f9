"""Data-parallel path on CPU: 2 ranks over gloo (the fake fabric; RCCL on the GPU box).

The DP contract (mfp/dp.py): shard the batch on axis 0, each rank's loss is a
mean over its shard, SUM-all-reduce the flat gradient, and fold 1/N + the L2 term + per-variable
clipnorm into the optimizer AFTER the reduction.  Result must equal the 1-rank step on the global
batch.  Gradients come from the oracle's torch restatement (the HIP kernels need a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _grads(ic, params, batch, masks, L, S):
    from oracle import torch_ref
    state = torch_ref.TrainState(params, l2=None, clipnorm=None, dtype=torch.float64)
    info, grads = torch_ref.loss_and_grads(state, ic, batch, batch, masks, L, maxlen=S)
    return info, grads


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "flex-dm_amd")]
    from oracle import np_ref
    from mfp import dp
    from mfp.data.spec import make_input_columns, synthetic_batch
    assert dp.init_from_env("gloo") == world and dp.rank() == rank and dp.world_size() == world
    ic = make_input_columns("rico")
    B, S, D, L = 4, 6, 16, 1
    params = np_ref.init_params(ic, D, L, seed=-7)
    batch = synthetic_batch(ic, B, S, seed=9, ragged=True)
    g = torch.Generator().manual_seed(3)
    keys = [k for k, c in ic.items() if c.get("is_sequence")]
    masks = {k: torch.rand(B, S, generator=g) < 0.6 for k in keys}
    shard = dp.shard_batch(batch)
    mshard = dp.shard_batch(masks)
    assert shard["left"].shape[0] == B // world
    info, grads = _grads(ic, params, shard, mshard, L, S)
    names = sorted(grads)
    flat = torch.cat([grads[n].reshape(-1) for n in names])
    dp.allreduce_gradients(flat)                      # SUM; the optimizer applies grad_scale = 1/N
    flat = flat / world
    sums = torch.tensor([[float(info["losses"][k]), float(info["scores"][k + "_score_num"]),
                          float(info["scores"][k + "_score_den"])] for k in keys], dtype=torch.float64)
    red = dp.allreduce_sums(sums)
    w = torch.full((5,), float(rank))
    dp.broadcast_parameters(w, src=0)
    if rank == 0:
        q.put((names, flat.numpy(), red.numpy(), w.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_global_batch_step():
    from oracle import np_ref
    from mfp.data.spec import make_input_columns, synthetic_batch
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    names, flat, red, w = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ic = make_input_columns("rico")
    B, S, D, L = 4, 6, 16, 1
    params = np_ref.init_params(ic, D, L, seed=-7)
    batch = synthetic_batch(ic, B, S, seed=9, ragged=True)
    g = torch.Generator().manual_seed(3)
    keys = [k for k, c in ic.items() if c.get("is_sequence")]
    masks = {k: torch.rand(B, S, generator=g) < 0.6 for k in keys}
    info, grads = _grads(ic, params, batch, masks, L, S)
    want = np.concatenate([grads[n].reshape(-1).numpy() for n in names])
    np.testing.assert_allclose(flat, want, rtol=1e-9, atol=1e-12)      # tolerance 1e-5 asked; f64 gives 1e-9
    for i, k in enumerate(keys):
        assert abs(red[i, 0] - float(info["losses"][k])) < 1e-9
        assert abs(red[i, 1] - float(info["scores"][k + "_score_num"])) < 1e-9
        assert abs(red[i, 2] - float(info["scores"][k + "_score_den"])) < 1e-9
    assert (w == 0).all()


def test_single_process_helpers_are_noops():
    from mfp import dp
    t = torch.ones(3)
    assert dp.world_size() == 1 and dp.rank() == 0
    assert dp.allreduce_gradients(t) is None and torch.equal(dp.allreduce_sums(t.view(1, 3)), t.view(1, 3))
    assert dp.shard_batch({"a": t})["a"] is t


def _eval_worker(rank, world, port, q, B):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "flex-dm_amd")]
    from oracle import np_ref, torch_ref
    from mfp import dp
    from mfp.data.spec import make_input_columns, synthetic_batch
    dp.init_from_env("gloo")
    ic = make_input_columns("rico")
    S, D, L = 6, 16, 1
    params = torch_ref.to_torch(np_ref.init_params(ic, D, L, seed=-7), torch.float64, requires_grad=False)
    batch = synthetic_batch(ic, B, S, seed=9, ragged=True)
    g = torch.Generator().manual_seed(3)
    keys = [k for k, c in ic.items() if c.get("is_sequence")]
    masks = {k: torch.rand(B, S, generator=g) < 0.6 for k in keys}
    shard, mshard = dp.shard_batch(batch, even=False), dp.shard_batch(masks, even=False)
    bl = shard["length"].shape[0]
    sums = None
    if bl > 0:
        out = torch_ref.model_fwd(params, ic, shard, L, maxlen=S)
        _, losses, scores, _ = torch_ref.loss_layer(ic, shard, out, mshard, S)
        sums = torch.tensor([[float(losses[k]), float(scores[k + "_score_num"]), float(scores[k + "_score_den"])]
                             for k in keys], dtype=torch.float64)
    red = dp.allreduce_eval_sums(sums, bl, len(keys), "cpu")
    seeds = dp.rank_seed(5)
    if rank == 0:
        q.put((red.numpy(), bl))
    q.put(("seed", rank, seeds))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 5), (3, 2)])
def test_evaluation_sums_over_uneven_shards_equal_the_global_batch(world, B):
    """evaluate() deals a ragged (last) batch out unevenly -- a shard may even be empty -- and
    all-reduces loss sums with the document counts: every rank must end up with the metrics of the
    whole batch (rank 0 used to report 1/N of the data)."""
    from oracle import np_ref, torch_ref
    from mfp import dp
    from mfp.data.spec import make_input_columns, synthetic_batch
    assert [dp.shard_bounds(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [dp.shard_bounds(2, r, 3) for r in range(3)] == [(0, 1), (1, 2), (2, 2)]
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, q, B)) for r in range(world)]
    for p in procs:
        p.start()
    got, seeds = None, {}
    for _ in range(world + 1):
        item = q.get(timeout=240)
        if isinstance(item[0], str):
            seeds[item[1]] = item[2]
        else:
            got = item[0]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert seeds[0] == 5 and len(set(seeds.values())) == world      # rank-dependent masking/dropout streams
    ic = make_input_columns("rico")
    S, D, L = 6, 16, 1
    params = torch_ref.to_torch(np_ref.init_params(ic, D, L, seed=-7), torch.float64, requires_grad=False)
    batch = synthetic_batch(ic, B, S, seed=9, ragged=True)
    g = torch.Generator().manual_seed(3)
    keys = [k for k, c in ic.items() if c.get("is_sequence")]
    masks = {k: torch.rand(B, S, generator=g) < 0.6 for k in keys}
    out = torch_ref.model_fwd(params, ic, batch, L, maxlen=S)
    _, losses, scores, _ = torch_ref.loss_layer(ic, batch, out, masks, S)
    for i, k in enumerate(keys):
        assert abs(got[i, 0] - float(losses[k])) < 1e-5 * max(1.0, abs(float(losses[k]))), k
        assert abs(got[i, 1] - float(scores[k + "_score_num"])) < 1e-5 and abs(got[i, 2] - float(scores[k + "_score_den"])) < 1e-6


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "flex-dm_amd")]
    from mfp import dp
    assert dp.init_from_env("gloo") == world
    gen = torch.Generator().manual_seed(100 + rank)          # every rank its own gradients
    n = 10007
    g = torch.randn(n, generator=gen)
    ref = g.clone()
    dp.allreduce_gradients(ref)                               # the single all-reduce of the whole buffer
    # the bucketed form: descending cut offsets as the backward pass completes them, asynchronous launches
    offsets = [9000, 4097, 4096, 13]
    slices = dp.bucket_slices(offsets, n)
    out = {}
    for name in ("f32", "bf16"):
        buf = g.clone()
        red = dp.BucketReducer(name)
        for sl in slices:
            red.launch(buf[sl])
        red.finish()
        out[name] = buf
    gathered = [torch.empty_like(out["bf16"]) for _ in range(world)]
    dist.all_gather(gathered, out["bf16"])
    if rank == 0:
        q.put((ref.numpy(), out["f32"].numpy(), out["bf16"].numpy(), [t.numpy() for t in gathered],
               [(s.start, s.stop) for s in slices]))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_single_allreduce():
    """mfp.dp.BucketReducer over per-segment buckets (what the captured data-parallel step launches behind every
    segment of its backward pass) against ONE all-reduce of the flat buffer: bit for bit in f32 (a sum all-reduce is
    element-wise); the bf16 carrier (MFP_DP_GRAD_DTYPE=bf16) gives every rank the SAME values, within bf16 rounding
    of the f32 sums."""
    from mfp import dp
    assert dp.bucket_cut_blocks(4, "blocks") == [3, 2, 1] and dp.bucket_cut_blocks(4, "halves") == [2]
    assert dp.bucket_cut_blocks(1, "blocks") == [] and dp.bucket_cut_blocks(8, "none") == []
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ref, f32, bf16, gathered, slices = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert slices[0] == (9000, 10007) and slices[-1] == (0, 13) and sum(b - a for a, b in slices) == 10007
    assert np.array_equal(ref, f32)
    assert all(np.array_equal(gathered[0], t) for t in gathered)
    assert np.allclose(bf16, ref, rtol=2 ** -7, atol=1e-2)


def _plan_worker(rank, world, port, q):
    """Three data-parallel train steps of the oracle's model on a rank-local shard under three all-reduce plans."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "flex-dm_amd")]
    from oracle import np_ref, torch_ref
    from mfp import dp
    from mfp.data.spec import make_input_columns, synthetic_batch
    assert dp.init_from_env("gloo") == world
    ic = make_input_columns("rico")
    B, S, D, L = 4, 6, 16, 2
    params = np_ref.init_params(ic, D, L, seed=-7)
    keys = [k for k, c in ic.items() if c.get("is_sequence")]
    names = sorted(torch_ref.TrainState(params).p)
    sizes = [int(np.prod(params[n].shape)) for n in names]
    numel = sum(sizes)
    result = {}
    for plan, carrier in (("blocks", "f32"), ("none", "f32"), ("blocks", "bf16")):
        state = torch_ref.TrainState(params, lr=1e-2, l2=1e-2, clipnorm=1.0, dtype=torch.float64)
        cuts = [numel * 3 // 4, numel // 2, numel // 5] if plan == "blocks" else []      # descending, like the backward pass
        for step in range(3):
            batch = dp.shard_batch(synthetic_batch(ic, B, S, seed=20 + step, ragged=True))
            g = torch.Generator().manual_seed(step)
            masks = dp.shard_batch({k: torch.rand(B, S, generator=g) < 0.6 for k in keys})
            loc = torch_ref.TrainState({k: v.detach().numpy() for k, v in state.p.items()}, l2=None, clipnorm=None, dtype=torch.float64)
            _, grads = torch_ref.loss_and_grads(loc, ic, batch, batch, masks, L, maxlen=S)
            flat = torch.cat([grads[n].reshape(-1) for n in names]).float()
            red = dp.BucketReducer(carrier)
            for sl in dp.bucket_slices(cuts, numel):
                red.launch(flat[sl])
            red.finish()
            flat = flat.double() / world
            avg, pos = {}, 0
            for n, sz in zip(names, sizes):
                avg[n] = flat[pos:pos + sz].reshape(state.p[n].shape) + 2.0 * state.l2 * state.p[n].detach() * (0.0 if n.split("/")[-1] in ("gamma", "beta") else 1.0)
                pos += sz
            torch_ref.apply_gradients(state, avg)
        result[(plan, carrier)] = torch.cat([state.p[n].detach().reshape(-1) for n in names])
    gathered = {}
    for key, w in result.items():
        parts = [torch.empty_like(w) for _ in range(world)]
        dist.all_gather(parts, w)
        gathered[key] = [p.numpy() for p in parts]
    if rank == 0:
        w0 = torch.cat([torch.as_tensor(np.asarray(params[n], dtype=np.float64)).reshape(-1) for n in names]).numpy()
        q.put((w0, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_plans_give_the_same_parameters_after_three_steps():
    """the default plan (per-block buckets, f32), the MFP_DP_BUCKETS=none fallback (one all-reduce)
    and the bf16 carrier (MFP_DP_GRAD_DTYPE=bf16) after 3 data-parallel steps: replicas bit-identical under every plan,
    bucketed == single all-reduce bit for bit, and the bf16 carrier's parameters within bf16 rounding of the gradients
    (the update direction agrees; Keras Adam's first steps are sign-like, so single entries may differ by ~lr)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    w0, gathered = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for key, parts in gathered.items():
        assert all(np.array_equal(parts[0], t) for t in parts), key            # replicas in sync
    ref = gathered[("blocks", "f32")][0]
    assert np.array_equal(ref, gathered[("none", "f32")][0])
    b16 = gathered[("blocks", "bf16")][0]
    d_ref, d_b16 = ref - w0, b16 - w0
    cos = float(d_ref @ d_b16 / (np.linalg.norm(d_ref) * np.linalg.norm(d_b16)))
    assert cos > 0.995, cos
    assert np.abs(b16 - ref).max() <= 2 * 3 * 1e-2        # never further apart than the steps themselves


def test_describe_plan_covers_every_parameter_once():
    from mfp import dp
    from mfp.data.spec import make_input_columns
    from mfp.models.params import ModelLayout
    lay = ModelLayout(make_input_columns("crello"), 256, 4)
    for mode, nb in (("blocks", 4), ("halves", 2), ("none", 1)):
        os.environ["MFP_DP_BUCKETS"] = mode
        try:
            plan = dp.describe_plan(lay)
        finally:
            os.environ.pop("MFP_DP_BUCKETS")
        assert len(plan["plan"]) == nb and sum(b["params"] for b in plan["plan"]) == lay.numel
        assert plan["bytes_per_step"] == 4 * lay.numel and plan["carrier"] == "f32"
    assert dp.describe_plan(lay, graphed=False)["plan"][0]["params"] == lay.numel


def test_plan_switches_are_validated_at_init(monkeypatch):
    """MFP_DP_BUCKETS / MFP_DP_GRAD_DTYPE are read at capture time; init_from_env rejects a bad value up front."""
    from mfp import dp
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("MFP_DP_BUCKETS", "quarters")
    with pytest.raises(ValueError):
        dp.init_from_env("gloo")
    monkeypatch.setenv("MFP_DP_BUCKETS", "blocks")
    monkeypatch.setenv("MFP_DP_GRAD_DTYPE", "fp8")
    with pytest.raises(ValueError):
        dp.init_from_env("gloo")
