"""``train(args)`` -- the reference's trainer glue (src/mfp/mfp/train.py:16-97) on this engine:
seeds, ``job_dir/args.json``, DataSpec datasets, MFP model, optional warm start, Adam(lr,
clipnorm=1.0), fit / evaluate (prints ``metric value`` lines), ``checkpoints/{best,final}.ckpt``.
Data-parallel when launched under torchrun (one process per GPU, RCCL)."""
import json
import logging
import os
import random

import numpy as np
import torch

from mfp import dp
from mfp.data import DataSpec
from mfp.helpers.callbacks import get_callbacks
from mfp.models.mfp import MFP

logger = logging.getLogger(__name__)


def fused_path_hint(dtype, latent_dim, seq_len, docs_per_rank):
    """One line when the run will NOT take the document-tile / activation-stationary kernels (mfp/hip/functions.py), with
    the flag that puts it there; None when it will.  The datasets' sequences are at most 51 positions long (data/*-spec.yml:
    length <= 50), so ``--seq_len 64`` (two documents per 128-row tile) costs 1.25x the padding of the longest batch and runs
    the fused kernels; without ``--seq_len`` every batch has its own length and is stepped on the generic kernels."""
    if dtype not in ("bf16", "fp8"):
        return None      # (the f32 parity path has no fused kernels)
    if latent_dim == 512:
        return None      # csrc/block_d512.hip takes any sequence length
    if latent_dim != 256:
        return ("latent_dim %d runs on the generic tile kernels (the fused kernels are built for --latent_dim 256 and 512)" % latent_dim)
    if seq_len == 128 or (seq_len == 64 and docs_per_rank % 2 == 0):
        return None
    if seq_len == 64:
        return "--seq_len 64 needs an even number of documents per GPU for the document-tile kernels (got %d)" % docs_per_rank
    return ("--seq_len %s falls off the document-tile kernels (generic attention / projection launches, ~1.3x slower per element); "
            "use --seq_len 64 (two documents per tile; sequences are at most 51 positions long) or --seq_len 128" % seq_len)


def train(args):
    logger.info(f"torch version {torch.__version__}")
    world = dp.init_from_env()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = args.device
    if device == "cuda":
        local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        device = "cuda:%d" % local_rank
    seed = args.seed
    torch.manual_seed(seed + dp.rank())
    np.random.seed(seed)
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)

    os.makedirs(args.job_dir, exist_ok=True)
    if dp.rank() == 0:
        with open(os.path.join(args.job_dir, "args.json"), "w") as file_obj:
            json.dump(vars(args), file_obj, indent=2)
    checkpoint_dir = os.path.join(args.job_dir, "checkpoints")
    checkpoint_path = os.path.join(checkpoint_dir, "best.ckpt")

    hint = fused_path_hint(args.dtype, args.latent_dim, args.seq_len, args.batch_size // max(world, 1))
    if hint and dp.rank() == 0:
        print("mfp.train: " + hint, flush=True)
    dataspec = DataSpec(args.dataset_name, args.data_dir, batch_size=args.batch_size,
                        seq_len=args.seq_len, device=device)
    train_dataset = dataspec.make_dataset("train", shuffle=True, repeat=True, cache=True)
    val_dataset = dataspec.make_dataset("val", cache=True)
    test_dataset = dataspec.make_dataset("test", cache=True)

    input_columns = dataspec.make_input_columns()
    model = MFP(input_columns, num_blocks=args.num_blocks, block_type=args.block_type,
                masking_method=args.masking_method, seq_type=args.seq_type, arch_type=args.arch_type,
                context=args.context, latent_dim=args.latent_dim, dropout=args.dropout, l2=args.l2,
                input_dtype=args.input_dtype, dtype=args.dtype, device=device, seed=seed)
    if args.weights:
        logger.info("Loading %s" % args.weights)
        model.load_weights(args.weights)
    dp.broadcast_parameters(model.model.store.w)
    model.model.store.refresh_shadow()

    model.compile(learning_rate=args.learning_rate, clipnorm=1.0, run_eagerly=True)
    model.fit(train_dataset, steps_per_epoch=dataspec.steps_per_epoch("train"), epochs=args.num_epochs,
              validation_data=val_dataset, validation_steps=dataspec.steps_per_epoch("val"),
              validation_freq=min(args.validation_freq, args.num_epochs),
              callbacks=get_callbacks(args, dataspec, checkpoint_path), verbose=args.verbose,
              use_graph=args.use_graph)

    results = model.evaluate(test_dataset, batch_size=args.batch_size)
    if dp.rank() == 0:
        for k, v in zip(model.metrics_names, results):
            print(k, v)
    model_path = os.path.join(args.job_dir, "checkpoints", "final.ckpt")
    logger.info("Saving %s" % model_path)
    model.save_weights(model_path)
    return model
