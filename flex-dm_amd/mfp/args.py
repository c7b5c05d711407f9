"""Command-line flags of ``python -m mfp`` -- names and defaults are the drop-in surface of the
reference (src/mfp/mfp/args.py:6-128; note ``--job-dir`` has a hyphen, every other flag
underscores).  Additive flags of this engine: ``--dtype``, ``--device``, ``--seq_len``,
``--use_graph``.
"""
import argparse

DATASET_NAMES = ["rico", "crello"]

# (flag, kwargs) in the reference's order
_BASE_FLAGS = [
    ("--dataset_name", dict(required=True, choices=DATASET_NAMES, help="Name of the dataset.")),
    ("--data_dir", dict(help="The GCS or local path of the data location ('synthetic[:S[:docs]]' "
                             "for synthetic batches).")),
    ("--weights", dict(default=None, type=str, help="Path to the initial model weight.")),
    ("--latent_dim", dict(default=256, type=int, help="Latent dimension.")),
    ("--num_blocks", dict(default=4, type=int, help="Number of stacked blocks in sequence encoder.")),
    ("--arch_type", dict(default="oneshot", help="Overall model type")),
    ("--block_type", dict(default="deepsvg", help="Stacked block type.")),
    ("--l2", dict(default=1e-2, type=float, help="Scalar coefficient for L2 regularization.")),
    ("--dropout", dict(default=0.1, type=float, help="Scalar ratio for dropout in transformer")),
    ("--masking_method", dict(type=str, default="random")),
    ("--seq_type", dict(type=str, default="default", choices=["default", "flat", "concat_enc"],
                        help="transformer's input is: element-wise feature (default), field-wise feature (flat)")),
    ("--log_level", dict(default="INFO", type=str)),
    ("--verbose", dict(default=2, type=int)),
    ("--seed", dict(default=0, type=int)),
    ("--mult", dict(default=1.0, type=float)),
    ("--context", dict(default=None)),
    ("--input_dtype", dict(type=str, default="set", choices=["set", "shuffled_set"])),
    ("--batch_size", dict(default=256, type=int)),
    # ---- additive (this engine)
    ("--dtype", dict(default="bf16", choices=["fp32", "bf16", "fp8"],
                     help="compute dtype of the HIP kernels (fp32 = exact-f32 MFMA parity path; fp8 = bf16 with "
                          "e4m3 QKV / FFN1 forward products)")),
    ("--device", dict(default="cuda", help="HIP device of this rank")),
    ("--seq_len", dict(default=None, type=int, help="padded sequence length of every batch.  Unset: bf16 / fp8 at --latent_dim 256 "
                       "pad each batch to 64 positions (128 if a document is longer) so that it runs on the document-tile kernels "
                       "(padding is inert); 0: pad each batch to its longest document as the reference does")),
    ("--use_graph", dict(action="store_true", help="capture the train step into hipGraphs")),
]

_TRAIN_FLAGS = [
    ("--job-dir", dict(required=True, help="The GCS or local path of logs and saved models.")),
    ("--num_epochs", dict(default=500, type=int, help="Number of epochs to train.")),
    ("--learning_rate", dict(default=1e-4, type=float, help="Base learning rate.")),
    ("--enable_profile", dict(dest="enable_profile", action="store_true",
                              help="Profile the 2nd train step: roctx range + torch.profiler trace under job_dir/logs.")),
    ("--validation_freq", dict(default=10, type=int, help="Validation frequency in terms of epochs.")),
]


class BaseArgs:
    def __init__(self):
        self.parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        for flag, kw in _BASE_FLAGS:
            self.parser.add_argument(flag, **kw)

    def parse_args(self, argv=None):
        return self.parser.parse_args(argv)


class TrainArgs(BaseArgs):
    def __init__(self):
        super().__init__()
        for flag, kw in _TRAIN_FLAGS:
            self.parser.add_argument(flag, **kw)

    def __call__(self, argv=None):
        return self.parser.parse_args(argv)
