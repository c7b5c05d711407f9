"""Training callbacks (reference helpers/callbacks.py:35-60): best-checkpoint on
``val_total_score`` (mode max, weights only), TerminateOnNaN, and -- for ``--enable_profile`` --
the counterpart of ``TensorBoard(profile_batch=2)`` (callbacks.py:44-48): the second train step
is wrapped in a roctx range (visible to ``rocprofv3 --marker-trace``) and traced with
``torch.profiler`` into ``job_dir/logs/profile_step2.*``.  TensorBoard scalars / HyperTune are
observability glue outside the hot path and are not provided."""
import logging
import math
import os

logger = logging.getLogger(__name__)


class ModelCheckpoint:
    def __init__(self, filepath, monitor="val_total_score", mode="max"):
        self.filepath, self.monitor = filepath, monitor
        self.best = -math.inf if mode == "max" else math.inf
        self.better = (lambda a, b: a > b) if mode == "max" else (lambda a, b: a < b)

    def on_epoch_end(self, epoch, logs, model):
        if self.monitor in logs and self.better(logs[self.monitor], self.best):
            self.best = logs[self.monitor]
            logger.info("Epoch %d: %s improved to %.5f, saving %s", epoch + 1, self.monitor, self.best,
                        self.filepath)
            model.save_weights(self.filepath)


class TerminateOnNaN:
    def on_epoch_end(self, epoch, logs, model):
        loss = logs.get("loss")
        if loss is not None and (math.isnan(loss) or math.isinf(loss)):
            logger.error("Epoch %d: invalid loss, terminating training", epoch + 1)
            model.stop_training = True


class ProfileStep:
    """``profile_batch=2`` of the reference's TensorBoard callback: profile the 2nd train step.

    * a roctx range ``mfp_train_step_<n>`` brackets the step (``rocprofv3 --marker-trace --kernel-trace
      -- python -m mfp ... --enable_profile`` attributes the kernels between the markers to it);
    * ``torch.profiler`` (roctracer) records the step's kernels; the chrome trace and a per-kernel table
      go to ``<log_dir>/profile_step<n>.trace.json`` / ``.kernels.txt``.
    The step is synchronised on both sides, so it is timed in isolation (as TensorBoard's profiler does)."""

    def __init__(self, log_dir: str, profile_batch: int = 2):
        self.log_dir, self.profile_batch = log_dir, profile_batch
        self._prof = None
        self.done = False

    def on_train_batch_begin(self, step: int):
        if self.done or step + 1 != self.profile_batch:
            return
        import torch
        torch.cuda.synchronize()
        try:
            from torch.profiler import ProfilerActivity, profile
            self._prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA])
            self._prof.__enter__()
        except Exception as exc:   # e.g. an external profiler already owns the tracer
            logger.warning("torch.profiler unavailable (%s); roctx range only", exc)
            self._prof = None
        try:
            torch.cuda.nvtx.range_push("mfp_train_step_%d" % self.profile_batch)   # roctxRangePush on ROCm
            self._range = True
        except Exception as exc:   # a build without roctx: the torch.profiler trace alone
            logger.warning("roctx range unavailable (%s)", exc)
            self._range = False

    def on_train_batch_end(self, step: int):
        if self.done or step + 1 != self.profile_batch:
            return
        import torch
        torch.cuda.synchronize()
        self.done = True
        if getattr(self, "_range", False):
            try:
                torch.cuda.nvtx.range_pop()
            except Exception as exc:
                logger.warning("roctx range_pop failed (%s)", exc)
        if self._prof is None:
            return
        prof, self._prof = self._prof, None
        try:
            prof.__exit__(None, None, None)
        except Exception as exc:
            logger.warning("could not stop the profiler: %s", exc)
            return
        self._prof = prof
        try:
            base = os.path.join(self.log_dir, "profile_step%d" % self.profile_batch)
            self._prof.export_chrome_trace(base + ".trace.json")
            with open(base + ".kernels.txt", "w") as f:
                f.write(self._prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=200))
            logger.info("profile of train step %d written to %s.*", self.profile_batch, base)
        except Exception as exc:
            logger.warning("could not export the profile: %s", exc)
        self._prof = None

    def on_epoch_end(self, epoch, logs, model):
        pass


def get_callbacks(args, dataspec, checkpoint_path: str):
    log_dir = os.path.join(args.job_dir, "logs")
    os.makedirs(log_dir, exist_ok=True)
    callbacks = [ModelCheckpoint(checkpoint_path), TerminateOnNaN()]
    if getattr(args, "enable_profile", False):
        callbacks.append(ProfileStep(log_dir, profile_batch=2))
    return callbacks
