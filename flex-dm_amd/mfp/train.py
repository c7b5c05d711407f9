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
    length <= 50), so 64 positions (two documents per 128-row tile) cost 1.25x the padding of the longest batch and run
    the fused kernels: that is what an unset ``--seq_len`` resolves to (default_seq_len); ``--seq_len 0`` keeps the reference's
    per-batch lengths, which are stepped on the generic kernels."""
    if dtype not in ("bf16", "fp8"):
        return None      # (the f32 parity path has no fused kernels)
    if latent_dim == 512:
        return None      # csrc/block_d512.hip takes any sequence length
    if latent_dim != 256:
        return ("latent_dim %d runs on the generic tile kernels (the fused kernels are built for --latent_dim 256 and 512)" % latent_dim)
    if isinstance(seq_len, (tuple, list)):      # the default: every batch padded to 64 (128) positions
        return None if docs_per_rank % 2 == 0 else ("batches of %d documents per GPU: 64-position documents go two to a tile, "
                                                    "an odd batch runs on the generic kernels" % docs_per_rank)
    if seq_len == 128 or (seq_len == 64 and docs_per_rank % 2 == 0):
        return None
    if seq_len == 64:
        return "--seq_len 64 needs an even number of documents per GPU for the document-tile kernels (got %d)" % docs_per_rank
    return ("--seq_len %s falls off the document-tile kernels (generic attention / projection launches, ~1.3x slower per element); "
            "use --seq_len 64 (two documents per tile; sequences are at most 51 positions long) or --seq_len 128" % seq_len)


def default_seq_len(dtype, latent_dim, seq_len):
    """What ``--seq_len`` resolves to.  Unset (None): on the bf16 / fp8 path at d_model 256 every batch is padded to 64 positions
    (128 when its longest document is longer; beyond that its own length) -- the reference pads a batch to its longest document
    (src/mfp/mfp/data/spec.py:255-276), which lands every batch on the generic kernels; padding is inert (masked keys, masked
    losses: tests/test_gpu_fullsize.py::test_padding_is_inert) and the datasets' documents are at most 51 positions long, so the
    reference's own command line (bin/train_mfp.sh:16-20) steps on the document-tile kernels.  ``--seq_len 0`` = the reference's
    ragged batches; any other value pads every batch to it."""
    if seq_len is None:
        return (64, 128) if dtype in ("bf16", "fp8") and latent_dim == 256 else None
    return None if seq_len == 0 else seq_len


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

    seq_len = default_seq_len(args.dtype, args.latent_dim, args.seq_len)
    hint = fused_path_hint(args.dtype, args.latent_dim, seq_len, args.batch_size // max(world, 1))
    if hint and dp.rank() == 0:
        print("mfp.train: " + hint, flush=True)
    dataspec = DataSpec(args.dataset_name, args.data_dir, batch_size=args.batch_size,
                        seq_len=seq_len, device=device)
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
