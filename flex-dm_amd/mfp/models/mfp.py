"""MFP task wrapper: task sampling, masking, model call, loss hookup, merge (reference
models/mfp.py) plus the Keras-like ``compile / fit / evaluate / save_weights / load_weights``
surface that reference train.py:67-97 and eval.py:169-172 use.

The compute (``self.model`` and the loss) runs in HIP kernels; this file is host logic.  What is
new relative to the reference: ``train_step`` (what Keras' ``Model.train_step`` did implicitly),
optional hipGraph capture of the whole step, and data-parallel gradient averaging (mfp.dp).
"""
from __future__ import annotations

import logging
import os
from typing import Dict, List, Optional

import torch

from mfp import dp
from mfp.data.spec import get_attribute_groups, get_dataset_name
from mfp.models.architecture.mask import get_seq_mask
from mfp.models.masking import (apply_token, elem_masking, feat_masking, filter_padding,
                                get_task_names, random_masking)
from mfp.models.metrics import (LossLayer, build_loss_keys, build_loss_sort, loss_key_names,
                                metrics_from_sums)
from mfp.models.fast_masking import FusedMasker
from mfp.models.model import Model
from mfp.models.tensor_utils import shuffle_inputs, sort_inputs
from mfp.optim import AdamKeras

logger = logging.getLogger(__name__)
logger.setLevel(logging.INFO)


def ops_hip():
    from mfp.hip import ops
    return ops


def get_task_probs(task_names: List[str], masking_method: str) -> List[float]:
    """reference mfp.py:34-43 (the Categorical's probabilities)."""
    used_names = masking_method.split("_")
    probs = [1.0 if name in used_names else 0.0 for name in task_names]
    probs_total = sum(probs)
    assert probs_total > 0.0
    probs = [p / probs_total for p in probs]
    logger.info([item for item in zip(task_names, probs)])
    return probs


def merge_inputs_and_prediction(inputs, input_columns, masks, prediction):
    """reference mfp.py:46-69."""
    for key, column in input_columns.items():
        if column.get("demo_only", False):
            continue
        if not column["is_sequence"]:
            prediction[key] = inputs[key]
        elif key not in masks.keys():
            continue
        elif column["type"] == "numerical":
            cond = masks[key][..., None]
            prediction[key] = torch.where(cond, prediction[key], inputs[key].to(prediction[key].dtype))
        else:
            gt = torch.nn.functional.one_hot(inputs[key].to(torch.int64), column["input_dim"])
            cond = masks[key][..., None, None]
            prediction[key] = torch.where(cond, prediction[key], gt.to(prediction[key].dtype))
    for key, column in input_columns.items():
        if column.get("demo_only", False) and key in inputs:
            prediction[key] = inputs[key]
    return prediction


def preprocess_for_test(inputs, input_columns, masks, tasks=None):
    """reference mfp.py:72-92."""
    S = inputs["left"].shape[1]
    seq_mask = get_seq_mask(inputs["length"], maxlen=S)
    filtered_inputs = filter_padding(inputs, input_columns, seq_mask)
    modified_inputs = {}
    for key, column in input_columns.items():
        if not column["is_sequence"]:
            modified_inputs[key] = filtered_inputs[key]
            continue
        modified_inputs[key] = apply_token(filtered_inputs[key], column, masks[key], "masked")
    if tasks is None:
        tasks = torch.zeros(inputs["left"].shape[0], device=inputs["left"].device)
    modified_inputs["task"] = tasks[..., None]
    return modified_inputs


def preprocess_for_train(inputs: Dict[str, torch.Tensor], input_columns: Dict, tasks: torch.Tensor,
                         is_autoreg: bool = False, input_dtype: str = "set",
                         active_tasks: Optional[List[int]] = None, generator=None):
    """reference mfp.py:95-138.  ``active_tasks`` (new, optional) lists the task ids that can
    occur (non-zero sampling probability); variants of other tasks are never selected by the
    ``where`` below, so they are not computed at all."""
    assert tasks.dim() == 1
    attribute_groups = get_attribute_groups(input_columns.keys())
    if is_autoreg or input_dtype == "shuffled_set":
        inputs = shuffle_inputs(inputs)
    elif input_dtype == "sorted_set":
        inputs = sort_inputs(inputs, input_columns)
    S = inputs["left"].shape[1]
    seq_mask = get_seq_mask(inputs["length"], maxlen=S)
    filtered_inputs = filter_padding(inputs, input_columns, seq_mask)

    def wanted(i):
        return active_tasks is None or i in active_tasks

    if wanted(0):
        modified_inputs, masks = random_masking(filtered_inputs, input_columns, seq_mask, generator=generator)
    else:  # task 0 never sampled: start from the unmodified inputs / all-False masks
        modified_inputs = dict(filtered_inputs)
        masks = {k: (torch.zeros_like(seq_mask) if c["is_sequence"] else None)
                 for k, c in input_columns.items()}
        masks = {k: v for k, v in masks.items() if v is not None}
    data = []
    data.append(elem_masking(filtered_inputs, input_columns, seq_mask, is_autoreg, generator)
                if wanted(1) else None)
    for gi, attribute_group in enumerate(attribute_groups.values()):
        data.append(feat_masking(filtered_inputs, input_columns, seq_mask, attribute_group)
                    if wanted(gi + 2) else None)
    for key in list(modified_inputs.keys()):
        for i, item in enumerate(data):
            if item is None:
                continue
            modified_inputs_tmp, masks_tmp = item
            cond = tasks == (i + 1)
            if input_columns[key]["is_sequence"]:
                cond = cond[..., None]
            modified_inputs[key] = torch.where(cond[..., None], modified_inputs_tmp[key], modified_inputs[key])
            if input_columns[key]["is_sequence"]:
                masks[key] = torch.where(cond, masks_tmp[key], masks[key])
    modified_inputs["task"] = tasks[..., None]
    return inputs, modified_inputs, masks


def iterative_decode(model, masks, inputs, input_columns, modified_inputs, num_iter):
    """MaskGIT-like decoding (reference mfp.py:141-207)."""
    masks = dict(masks)
    S = inputs["left"].shape[1]
    seq_mask = get_seq_mask(inputs["length"], maxlen=S)
    filtered_inputs = filter_padding(inputs, input_columns, seq_mask)
    categorical_keys = [k for k, v in input_columns.items()
                        if v["is_sequence"] and v.get("type", None) == "categorical"]
    num_masked = sum(masks[k].to(torch.int64).sum(-1) for k in categorical_keys)
    num_update_per_iter = torch.round(num_masked.double() / num_iter).to(torch.int64)
    final_outputs, outputs = None, None
    for i in range(num_iter):
        outputs = model(modified_inputs, training=False)
        outputs.pop("_flat_logits", None)
        if i == 0:
            final_outputs = dict(outputs)
        confidence = {
            k: torch.where(masks[k], torch.softmax(outputs[k], dim=-1).max(dim=-1).values.mean(dim=-1),
                           torch.zeros((), device=masks[k].device))
            for k in categorical_keys
        }
        confidence_sorted = torch.sort(torch.cat([confidence[k] for k in categorical_keys], dim=-1),
                                       dim=-1, descending=True).values
        idx = num_update_per_iter.clamp(max=confidence_sorted.shape[1] - 1)
        threshold = torch.gather(confidence_sorted, 1, idx[:, None])[:, 0]
        for key in categorical_keys:
            pred = outputs[key].argmax(dim=-1).to(filtered_inputs[key].dtype)
            update_field = (confidence[key] >= threshold[:, None]) & (confidence[key] > 0)
            filtered_inputs[key] = torch.where(update_field[:, :, None], pred, filtered_inputs[key])
            masks[key] = torch.where(masks[key] == update_field, torch.zeros_like(masks[key]), masks[key])
            if i > 0:
                final_outputs[key] = torch.where(update_field[:, :, None, None], outputs[key], final_outputs[key])
        for key, column in input_columns.items():
            if column["is_sequence"]:
                modified_inputs[key] = apply_token(filtered_inputs[key], column, masks[key], "masked")
    for key in ["image_embedding", "text_embedding"]:
        if key in outputs:
            final_outputs[key] = outputs[key]
    return final_outputs


class MFP:
    """MFP trainer (reference mfp.py:210-347)."""

    def __init__(self, input_columns: Dict, num_blocks: int = 4, block_type: str = "deepsvg",
                 masking_method: str = "random", seq_type: str = "default", arch_type: str = "oneshot",
                 context: Optional[str] = None, input_dtype: str = "set", name: str = "mfp",
                 use_elemwise_noise: bool = False, **kwargs):
        assert arch_type == "oneshot"
        if seq_type != "default":
            raise NotImplementedError("seq_type=%r: only 'default' is on the MFP hot path" % seq_type)
        self.name = name
        self.arch_type, self.context, self.input_dtype = arch_type, context, input_dtype
        self._all_input_columns = input_columns
        self.input_columns = {k: v for (k, v) in input_columns.items() if not v.get("demo_only", False)}
        self.is_autoreg = False
        kwargs.pop("kl", None)
        self.model = Model(input_columns=input_columns, num_blocks=num_blocks, block_type=block_type,
                           context=context, input_dtype=input_dtype,
                           use_elemwise_noise=use_elemwise_noise, **kwargs)
        self.loss_layer = LossLayer(input_columns, model_layout=self.model.layout)
        self.task_names = get_task_names(input_columns)
        self.task_probs = get_task_probs(self.task_names, masking_method)
        self._active_tasks = [i for i, p in enumerate(self.task_probs) if p > 0.0]
        self._task_probs_dev = torch.tensor(self.task_probs, dtype=torch.float32, device=self.model.store.device)
        self.sort_pos = get_dataset_name(input_columns.keys()) == "rico"
        self.fast_masking = True   # fused HIP masking in train_step (same semantics, Philox stream)
        self._masker = FusedMasker(self.input_columns, self.model.layout, self.model.store, kwargs.get("seed", 0))
        self.optimizer: Optional[AdamKeras] = None
        self.stop_training = False
        self._graph = None
        self.last_sums = None

    # ------------------------------------------------------------------ reference call surface
    def sample_tasks(self, B: int) -> torch.Tensor:
        """tfp Categorical(logits=log probs).sample(B) (mfp.py:301) -> int32 (B,)."""
        return torch.multinomial(self._task_probs_dev, B, replacement=True).to(torch.int32)

    def __call__(self, inputs, training=False, demo_args=None):
        is_demo = True if demo_args else False
        B = inputs["left"].shape[0]
        tasks = self.sample_tasks(B)
        if is_demo:
            targets = inputs
            masks = demo_args["masks"]
            modified_inputs = preprocess_for_test(inputs, self.input_columns, masks, demo_args.get("tasks", tasks))
        else:
            targets, modified_inputs, masks = preprocess_for_train(
                inputs, self.input_columns, tasks, is_autoreg=self.is_autoreg,
                input_dtype=self.input_dtype, active_tasks=self._active_tasks)
        iter_decode = False
        if is_demo:
            num_iter = demo_args.get("num_iter", 1)
            iter_decode = num_iter > 1
        with torch.set_grad_enabled(False):
            if iter_decode:
                outputs = iterative_decode(self.model, masks, inputs, self.input_columns, modified_inputs, num_iter)
            else:
                outputs = self.model(modified_inputs, training)
        if not is_demo:
            if self.sort_pos:
                ind = self.task_names.index("pos")
                self.loss_layer((targets, outputs, masks), training, (tasks == ind))
            else:
                self.loss_layer((targets, outputs, masks), training)
        outputs.pop("_flat_logits", None)
        outputs = {k: v.clone() if torch.is_tensor(v) else v for k, v in outputs.items()}
        outputs = merge_inputs_and_prediction(inputs, self.input_columns, masks, outputs)
        outputs["tasks"] = tasks
        return outputs

    # ------------------------------------------------------------------ training
    def compile(self, optimizer=None, run_eagerly=True, learning_rate: float = 1e-4, clipnorm: float = 1.0,
                **kwargs):
        """``compile(optimizer=Adam(lr, clipnorm=1.0), run_eagerly=True)`` (train.py:71-77);
        ``compile(optimizer="adam")`` as eval.py:170 does before load_weights."""
        if isinstance(optimizer, AdamKeras):
            self.optimizer = optimizer
        else:
            self.optimizer = AdamKeras(self.model.store, learning_rate=learning_rate, clipnorm=clipnorm)
        self.model.step_ptr = self.optimizer.step_t

    def _forward(self, batch: Dict[str, torch.Tensor]):
        """masking -> encoder -> blocks -> heads + losses.  Returns (loss, sums, ctx or None).

        On the fused train path ``loss`` is only the ROOT of the backward pass (a zero scalar: nothing on the device
        needs the total, ``metrics_dict`` adds ``sums[:, 0]`` on the host) and ``sums`` is filled by the reduction launch
        at the END of the backward pass (``StepCtx.flush_ln_jobs``): read it after ``backward()``, never between
        ``_forward`` and ``backward``."""
        B = batch["left"].shape[0]
        ctx = None
        nvalid = sums_flat = None
        if self.fast_masking and self.input_dtype == "set" and self.model.store.device.type == "cuda":
            length = batch["length"].reshape(-1)
            if length.dtype == torch.int32 and length.is_contiguous():
                # the head of the step in ONE launch on its counter-based stream: task ids (torch.multinomial is ~8
                # tiny kernels), nvalid = length + 1, and the loss accumulators zeroed
                # (a captured step writes into a buffer that lives OUTSIDE the graphs' shared pool: capture_train_step)
                sums_flat = self._sums_buf if getattr(self, "_sums_buf", None) is not None else torch.empty(
                    3 * len(loss_key_names(self._all_input_columns)) + 1, dtype=torch.float32, device=length.device)
                tasks, nvalid = ops_hip().step_prologue(self.task_probs, length, self._masker.seed, 1,
                                                        self.model.step_ptr, sums_flat)
            else:
                tasks = ops_hip().sample_tasks(self.task_probs, B, self._masker.seed, 1, self.model.step_ptr,
                                               self.model.store.device)
        else:
            tasks = self.sample_tasks(B)
        # RICO: documents drawn for the "pos" task are scored order-free (mfp.py:336-338)
        sorted_loss = self.sort_pos and self.task_names.index("pos") in self._active_tasks

        def loss_sort(targets):
            if not sorted_loss:
                return None
            return build_loss_sort(self._all_input_columns, self.model.layout.head_cols, targets,
                                   tasks == self.task_names.index("pos"))
        if self.fast_masking and self.input_dtype == "set":
            ctx = self.model.make_ctx(batch, True, nvalid=nvalid)
            ctx.tail["sums"] = sums_flat
            ctx.tail["want_logits"] = False      # the step returns the per-key sums; nobody reads the logits
            idx_all, codes, xs, masks = self._masker(batch, tasks, ctx.nvalid, ctx.B, ctx.S, self.model.step_ptr)
            keys = build_loss_keys(self._all_input_columns, self.model.layout.head_cols, batch, masks)
            cin = {"task": tasks[..., None], "length": batch["length"]} if self.context is not None else None
            loss, sums, _ = self.model.forward_loss(cin, keys, training=True, premasked=(idx_all, codes, xs), ctx=ctx,
                                                    loss_sort=loss_sort(batch))
        else:
            targets, modified_inputs, masks = preprocess_for_train(
                batch, self.input_columns, tasks, is_autoreg=self.is_autoreg, input_dtype=self.input_dtype,
                active_tasks=self._active_tasks)
            keys = build_loss_keys(self._all_input_columns, self.model.layout.head_cols, targets, masks)
            loss, sums, _ = self.model.forward_loss(modified_inputs, keys, training=True, loss_sort=loss_sort(targets))
        return loss, sums, ctx

    def _unit_grad(self, loss: torch.Tensor) -> torch.Tensor:
        """The root gradient as a resident tensor: autograd would otherwise launch a fill kernel for
        ones_like(loss) on the critical path of every step."""
        one = getattr(self, "_one", None)
        if one is None or one.device != loss.device or one.dtype != loss.dtype:
            one = self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
        return one

    def _join_sides(self):
        for side in self.model.side_streams:   # weight gradients run on the side streams
            torch.cuda.current_stream().wait_stream(side)

    def _forward_backward(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        loss, sums, _ = self._forward(batch)
        loss.backward(self._unit_grad(loss))
        self._join_sides()
        return sums

    def _apply(self):
        n = dp.world_size()
        dp.allreduce_gradients(self.model.store.g)
        self.optimizer.step(grad_scale=1.0 / n)

    def train_step(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """masking -> fwd -> losses -> bwd -> (all-reduce) -> clipnorm + L2 + Adam.
        Returns the device tensor ``sums [nkeys][3]`` (no host sync)."""
        assert self.optimizer is not None, "call compile() first"
        if self._graph is not None:
            return self._graph(batch)
        sums = self._forward_backward(batch)
        self._apply()
        self.last_sums = sums
        return sums

    def capture_train_step(self, example_batch: Dict[str, torch.Tensor], warmup: int = 2, resident: int = 1):
        """Capture the whole train step into hipGraphs (launch-bound inner loop: ~10^2 kernels of
        ~10 us).  With world_size > 1 the step is one graph per segment of the backward pass (cut at the block
        inputs, dp.bucket_cut_blocks) plus one for Adam, with the RCCL all-reduce of each segment's gradient bucket
        running under the next segment.

        ``resident`` > 1 (single-rank replay only): that many input buffer sets, each with its own capture of the same
        step over the same activation pool; ``self.static_batches[i]`` are the buffers a loader fills, the replay
        function runs the capture whose buffers it is handed, and ``self.train_steps_resident()`` runs ONE graph that steps
        through all of them in order (bench.py rotates them so that a step's 4 KB per element of
        inputs come from HBM, not from the infinity cache a single re-masked batch would sit in)."""
        assert self.optimizer is not None, "call compile() first"
        self.train_steps_resident = None      # (state of an earlier capture)
        self.static_batches = None
        if resident > 1 and dp.world_size() == 1:
            return self._capture_resident(example_batch, warmup, resident)
        static = {k: v.clone() for k, v in example_batch.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._forward_backward(static)
                self._apply()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        multi = dp.active()
        self.dp_graph_launches_per_step = 1
        if multi and dp.graph_mode() != "segments" and torch.distributed.get_backend() == "nccl":
            # ONE graph for the whole data-parallel step, collectives included (dp.graph_mode); a refused capture falls
            # back to the per-segment graphs below
            try:
                return self._capture_dp_one_graph(static, side)
            except Exception as exc:      # noqa: BLE001 -- whatever the capture raised: say so and use the segment graphs
                if dp.graph_mode() == "one":
                    raise
                import sys
                print("mfp: one-graph capture of the data-parallel step refused (%s: %s); per-segment graphs"
                      % (type(exc).__name__, exc), file=sys.stderr)
                torch.cuda.synchronize()
                with torch.cuda.stream(side):      # (a clean eager step: the interrupted capture left half a step behind)
                    self._forward_backward(static)
                    self._apply()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        segs = []           # (graph, gradient bucket completed by it) in backward order; N > 1 only
        g3 = None
        grad = self.model.store.g
        layout = self.model.layout
        # thread-local capture checks: with a process group alive its watchdog thread polls events,
        # which a "global"-mode capture would treat as an illegal call and abort
        mode = "thread_local" if multi else "global"
        # capture on the stream the warm-up ran on: the per-stream scratch (weight-gradient tickets, workspaces) then
        # exists already -- created inside the capture, the tickets' zero-fill became a kernel node of every replay
        if not multi:
            with torch.cuda.graph(g1, stream=side):
                static_sums = self._forward_backward(static)
                self._apply()
        else:
            # N > 1: the backward pass is cut at block inputs (dp.bucket_cut_blocks: by default every block).  Graph 0 =
            # forward + heads + the last block; every further graph one more segment of the backward pass.  The flat
            # gradient buffer is laid out encoder | block 0 | ... | block L-1 | heads, so what a segment completes is a
            # contiguous bucket: its RCCL all-reduce is launched ASYNCHRONOUSLY behind the segment and runs under the
            # next one; only the last bucket (block 0 + encoder) and Adam (its own graph) are exposed.
            with torch.cuda.graph(g1, stream=side, capture_error_mode=mode):
                loss, static_sums, ctx = self._forward(static)
                cut_blocks = [i for i in dp.bucket_cut_blocks(layout.L) if ctx is not None and i in ctx.cuts]
                if cut_blocks:
                    x_prev = ctx.cuts[cut_blocks[0]]
                    d_prev = torch.autograd.grad(loss, x_prev, grad_outputs=self._unit_grad(loss))[0]
                    ctx.flush_ln_jobs()   # the segment's LayerNorm gradients must be final before its all-reduce
                else:
                    loss.backward(self._unit_grad(loss))
                self._join_sides()
            slices = dp.bucket_slices([layout.block_offset(i) for i in cut_blocks], grad.numel())
            segs.append((g1, slices[0]))
            for k in range(1, len(cut_blocks) + 1):
                gk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gk, pool=g1.pool(), stream=side, capture_error_mode=mode):
                    if k < len(cut_blocks):
                        x_k = ctx.cuts[cut_blocks[k]]
                        d_k = torch.autograd.grad(x_prev, x_k, grad_outputs=d_prev)[0]
                        ctx.flush_ln_jobs()
                        x_prev, d_prev = x_k, d_k
                    else:
                        x_prev.backward(d_prev)
                    self._join_sides()
                segs.append((gk, slices[k]))
            g3 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g3, pool=g1.pool(), stream=side, capture_error_mode=mode):
                self.optimizer.step(grad_scale=1.0 / dp.world_size())
        reducer = dp.BucketReducer() if multi else None

        def replay(batch):
            if any(k in static and tuple(v.shape) != tuple(static[k].shape) for k, v in batch.items()):
                # a batch of another shape (ragged last batch, variable S without --seq_len) cannot go
                # through the captured graph: step it eagerly (same kernels, same semantics)
                sums = self._forward_backward(batch)
                self._apply()
                self.last_sums = sums
                return sums
            for k, v in batch.items():
                if k in static and v.data_ptr() != static[k].data_ptr():
                    static[k].copy_(v, non_blocking=True)
            if not multi:
                g1.replay()
            else:
                for gk, sl in segs:
                    gk.replay()
                    reducer.launch(grad[sl])      # overlaps the next segment's graph
                reducer.finish()
                g3.replay()
            self.last_sums = static_sums
            return static_sums

        self._graph = replay
        self._graph_objs = (g1, segs, g3, static, static_sums)
        self.static_batch = static   # a loader that writes the next batch here avoids the copy
        self.dp_graph_launches_per_step = len(segs) + 1 if multi else 1
        return replay

    def _capture_dp_one_graph(self, static, side):
        """The N > 1 step as ONE hipGraph: forward, the backward pass cut at the block inputs (dp.bucket_cut_blocks), each
        segment's gradient bucket all-reduced ASYNCHRONOUSLY (the collective is captured on the process group's communication
        stream: a branch of the graph that runs beside the next segment's kernels), the join, Adam.  One graph launch per
        step instead of L + 1 with eager collectives between them."""
        grad = self.model.store.g
        layout = self.model.layout
        g = torch.cuda.CUDAGraph()
        reducer = dp.BucketReducer()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            loss, static_sums, ctx = self._forward(static)
            cut_blocks = [i for i in dp.bucket_cut_blocks(layout.L) if ctx is not None and i in ctx.cuts]
            slices = dp.bucket_slices([layout.block_offset(i) for i in cut_blocks], grad.numel())
            if cut_blocks:
                x_prev = ctx.cuts[cut_blocks[0]]
                d_prev = torch.autograd.grad(loss, x_prev, grad_outputs=self._unit_grad(loss))[0]
                ctx.flush_ln_jobs()   # the segment's LayerNorm gradients must be final before its all-reduce
            else:
                loss.backward(self._unit_grad(loss))
            self._join_sides()
            reducer.launch(grad[slices[0]])
            for k in range(1, len(cut_blocks) + 1):
                if k < len(cut_blocks):
                    x_k = ctx.cuts[cut_blocks[k]]
                    d_k = torch.autograd.grad(x_prev, x_k, grad_outputs=d_prev)[0]
                    ctx.flush_ln_jobs()
                    x_prev, d_prev = x_k, d_k
                else:
                    x_prev.backward(d_prev)
                self._join_sides()
                reducer.launch(grad[slices[k]])
            reducer.finish()
            self.optimizer.step(grad_scale=1.0 / dp.world_size())

        def replay(batch):
            if any(k in static and tuple(v.shape) != tuple(static[k].shape) for k, v in batch.items()):
                sums = self._forward_backward(batch)      # (another shape: eagerly, as the per-segment form does)
                self._apply()
                self.last_sums = sums
                return sums
            for k, v in batch.items():
                if k in static and v.data_ptr() != static[k].data_ptr():
                    static[k].copy_(v, non_blocking=True)
            g.replay()
            self.last_sums = static_sums
            return static_sums

        self._graph = replay
        self._graph_objs = (g, [], None, static, static_sums)
        self.static_batch = static
        self.dp_graph_launches_per_step = 1
        return replay

    def _capture_resident(self, example_batch, warmup: int, resident: int):
        statics = [{k: v.clone() for k, v in example_batch.items()} for _ in range(resident)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._forward_backward(statics[0])
                self._apply()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # The graphs share one activation pool and are replayed in ANY order, so nothing a caller reads later may live in
        # that pool (a later capture's output could sit in a block an earlier graph uses for transients): every capture
        # writes its per-key sums into its own buffer allocated here, outside the pool.
        nflat = 3 * len(loss_key_names(self._all_input_columns)) + 1
        dev = self.model.store.device
        outs = [torch.zeros(nflat, dtype=torch.float32, device=dev) for _ in range(resident + 1)]

        def pinned(s_, buf):
            if s_.data_ptr() == buf.data_ptr():      # the fused train path: the step wrote into `buf` itself
                return s_
            view = buf[:s_.numel()].view(s_.shape)   # other paths: one small copy node at the end of the step
            view.copy_(s_)
            return view
        graphs, sums = [], []
        try:
            for i, static in enumerate(statics):
                g = torch.cuda.CUDAGraph()
                self._sums_buf = outs[i]
                with torch.cuda.graph(g, stream=side, pool=graphs[0].pool() if graphs else None):
                    s_ = self._forward_backward(static)
                    self._apply()
                    s_ = pinned(s_, outs[i])
                graphs.append(g)
                sums.append(s_)
            # ... and ONE graph that steps through all the buffer sets in order: a training loop that has `resident` batches
            # staged pays the graph-launch gap (~8-13 us of idle queue between two replays) once per `resident` steps
            gall = torch.cuda.CUDAGraph()
            self._sums_buf = outs[resident]
            with torch.cuda.graph(gall, stream=side, pool=graphs[0].pool()):
                for static in statics:
                    all_sums = self._forward_backward(static)
                    self._apply()
                all_sums = pinned(all_sums, outs[resident])
        finally:
            self._sums_buf = None

        def replay_all():
            gall.replay()
            self.last_sums = all_sums
            return all_sums

        self.train_steps_resident = replay_all
        by_ptr = {st["left"].data_ptr(): i for i, st in enumerate(statics)}

        def replay(batch):
            if any(k in statics[0] and tuple(v.shape) != tuple(statics[0][k].shape) for k, v in batch.items()):
                s_ = self._forward_backward(batch)
                self._apply()
                self.last_sums = s_
                return s_
            i = by_ptr.get(batch["left"].data_ptr(), 0)
            for k, v in batch.items():      # (a dict that reuses some static buffers with other fresh tensors: copy those)
                if k in statics[i] and v.data_ptr() != statics[i][k].data_ptr():
                    statics[i][k].copy_(v, non_blocking=True)
            graphs[i].replay()
            self.last_sums = sums[i]
            return sums[i]

        self._graph = replay
        self._graph_objs = (graphs, [], None, statics, sums)
        self.static_batch = statics[0]
        self.static_batches = statics
        return replay

    def test_step(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Keras test_step: MFP.call(training=False) still re-masks (is_demo False).  Returns the
        LossLayer sums ``[nkeys][3]`` = (loss: mean over the batch's documents, score numerator,
        score denominator) of this batch (metrics.py:265-288)."""
        with torch.no_grad():
            B = batch["left"].shape[0]
            tasks = self.sample_tasks(B)
            targets, modified_inputs, masks = preprocess_for_train(
                batch, self.input_columns, tasks, input_dtype=self.input_dtype, active_tasks=self._active_tasks)
            outputs = self.model(modified_inputs, training=False)
            if self.sort_pos:                                            # mfp.py:336-338
                self.loss_layer((targets, outputs, masks), False, tasks == self.task_names.index("pos"))
            else:
                self.loss_layer((targets, outputs, masks), False)
            return self.loss_layer.sums

    @property
    def metrics_names(self) -> List[str]:
        keys = loss_key_names(self._all_input_columns)
        return ["loss"] + [k + "_score" for k in keys] + [k + "_loss" for k in keys] + ["total_score"]

    def metrics_dict(self, sums: torch.Tensor, include_reg: bool = True) -> Dict[str, float]:
        sums = dp.allreduce_sums(sums)
        losses, scores, metrics = metrics_from_sums(self._all_input_columns, sums)
        out = {k: float(v) for k, v in metrics.items()}
        loss = float(sums[:, 0].sum())
        if include_reg and self.optimizer is not None:
            loss += float(self.optimizer.reg_loss())
        out["loss"] = loss
        return out

    def fit(self, dataset, steps_per_epoch=None, epochs=1, validation_data=None, validation_steps=None,
            validation_freq=1, callbacks=None, verbose=2, use_graph: bool = False):
        callbacks = callbacks or []
        history = []
        global_step = 0
        it = iter(dataset)
        for epoch in range(epochs):
            acc, n = None, 0
            for _ in range(steps_per_epoch or len(dataset)):
                try:
                    batch = next(it)
                except StopIteration:
                    it = iter(dataset)
                    batch = next(it)
                batch = dp.shard_batch(batch)
                if use_graph and self._graph is None:
                    self.capture_train_step(batch)
                for cb in callbacks:
                    if hasattr(cb, "on_train_batch_begin"):
                        cb.on_train_batch_begin(global_step)
                sums = self.train_step(batch)
                for cb in callbacks:
                    if hasattr(cb, "on_train_batch_end"):
                        cb.on_train_batch_end(global_step)
                global_step += 1
                acc = sums.clone() if acc is None else acc + sums
                n += 1
            logs = self.metrics_dict(_mean_sums(acc, n))
            if validation_data is not None and (epoch + 1) % max(1, validation_freq) == 0:
                val = self.evaluate(validation_data, steps=validation_steps, return_dict=True)
                logs.update({"val_" + k: v for k, v in val.items()})
            history.append(logs)
            if verbose and dp.rank() == 0:
                print("Epoch %d/%d - " % (epoch + 1, epochs) +
                      " - ".join("%s: %.4f" % (k, v) for k, v in logs.items()
                                 if k in ("loss", "total_score", "val_loss", "val_total_score")), flush=True)
            for cb in callbacks:
                cb.on_epoch_end(epoch, logs, self)
            if self.stop_training:
                break
        return history

    def evaluate(self, dataset, batch_size=None, steps=None, return_dict=False):
        """Keras ``evaluate``: every metric is the mean over batches of its per-batch value
        (``add_metric`` aggregation "mean").  Data-parallel: each rank scores its shard of the batch
        and the sums are all-reduced with the document counts BEFORE normalising, so every rank
        reports the metrics of the whole batch (ragged last batches are dealt out unevenly)."""
        keys = loss_key_names(self._all_input_columns)
        acc, n = None, 0
        for i, batch in enumerate(dataset):
            if steps is not None and i >= steps:
                break
            shard = dp.shard_batch(batch, even=False)
            b_local = int(shard["length"].shape[0])
            s = self.test_step(shard) if b_local > 0 else None
            s = dp.allreduce_eval_sums(s, b_local, len(keys), self.model.store.device)
            _, _, m = metrics_from_sums(self._all_input_columns, s)
            row = torch.stack([torch.stack([m[k + "_loss"], m[k + "_score"]]) for k in keys])
            acc = row.clone() if acc is None else acc + row
            n += 1
        res = _metrics_from_eval(self._all_input_columns, acc, n)
        if return_dict:
            return res
        return [res[k] for k in self.metrics_names]

    # ------------------------------------------------------------------ checkpoints
    def save_weights(self, path: str):
        """Own format (safetensors) under the reference's file layout
        ``job_dir/checkpoints/{best,final}.ckpt*`` (train.py:34-35,95-97)."""
        from safetensors.torch import save_file
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        if dp.rank() == 0:
            save_file(dict(self.model.store.state_dict()), path + ".safetensors")

    def load_weights(self, path: str, name_map: Optional[Dict[str, str]] = None):
        """``model.load_weights(path)`` (train.py:67-69, eval.py:169-172): ``path`` is either a
        checkpoint written by :meth:`save_weights` (``<path>.safetensors``) or the prefix of a
        TensorFlow checkpoint of the reference (``<path>.index`` + ``<path>.data-*``), read
        without TensorFlow by ``mfp.data.tf_checkpoint``."""
        if os.path.exists(path + ".index"):
            from mfp.data.tf_checkpoint import read_state_dict
            expected = {k: tuple(v.shape) for k, v in self.model.store.state_dict().items()}
            self.model.store.load_state_dict(read_state_dict(path, expected, name_map))
            return self
        from safetensors.torch import load_file
        p = path if path.endswith(".safetensors") else path + ".safetensors"
        if not os.path.exists(p):
            raise FileNotFoundError("neither %s nor %s.index exists" % (p, path))
        self.model.store.load_state_dict(load_file(p))
        return self


def _mean_sums(acc: torch.Tensor, n: int) -> torch.Tensor:
    out = acc.clone()
    out[:, 0] /= max(n, 1)   # per-step batch means -> epoch mean; num/den stay sums
    return out


def _metrics_from_eval(input_columns, acc, n):
    keys = loss_key_names(input_columns)
    out = {}
    if acc is None:
        return {k: float("nan") for k in ["loss"] + [x + "_score" for x in keys] + [x + "_loss" for x in keys] + ["total_score"]}
    mean = acc / max(n, 1)
    total = 0.0
    for i, k in enumerate(keys):
        out[k + "_loss"] = float(mean[i, 0])
        out[k + "_score"] = float(mean[i, 1])
        total += float(mean[i, 1])
    out["loss"] = float(mean[:, 0].sum())
    out["total_score"] = total / len(input_columns)
    return out
