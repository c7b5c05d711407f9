"""Training callbacks (reference helpers/callbacks.py:35-60): best-checkpoint on
``val_total_score`` (mode max, weights only) and TerminateOnNaN.  TensorBoard / HyperTune are
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


def get_callbacks(args, dataspec, checkpoint_path: str):
    os.makedirs(os.path.join(args.job_dir, "logs"), exist_ok=True)
    return [ModelCheckpoint(checkpoint_path), TerminateOnNaN()]
