#!/usr/bin/env python
"""Per-task evaluation over the test split -- drop-in for the reference's ``eval.py`` (same flags,
same ``job_dir/args.json`` + ``checkpoints/best.ckpt`` inputs, same printed/CSV result dict), with
the model call running on the HIP kernels.

Differences from the reference, both documented in SURVEY.md section 3.3:
* ``--task_mode random`` works (the reference passes ``replace_prob/unchange_prob`` that its
  ``random_masking`` does not accept: eval.py:59-65 vs masking.py:227-231 -> TypeError);
* ``--task_mode elem|random`` works (the reference reads an undefined ``group_name``:
  eval.py:99 -> NameError); the task id sent with ``context="id"`` is that of the mode itself.
The RICO position-sorted score (``sort_flag``, reference metrics.py:180-211) runs on the device
(``mfp_sort_positions`` + ``mfp_loss_fwd_bwd_sorted``) like the train step's.
"""
import argparse
import csv
import json
import logging
import os
import random
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "flex-dm_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mfp.data import DataSpec  # noqa: E402
from mfp.data.spec import get_attribute_groups, get_dataset_name  # noqa: E402
from mfp.models.architecture.mask import get_seq_mask  # noqa: E402
from mfp.models.masking import get_initial_masks, get_task_names, random_masking  # noqa: E402
from mfp.models.metrics import LossLayer  # noqa: E402
from mfp.models.mfp import MFP  # noqa: E402

logger = logging.getLogger(__name__)
logging.basicConfig(level=logging.ERROR)

seed = 0
torch.manual_seed(seed)
np.random.seed(seed)
random.seed(seed)
os.environ["PYTHONHASHSEED"] = str(seed)


def evaluate(args, model, dataset, input_columns, group):
    group_name, group_keys = group if group else (args.task_mode, [])
    logger.info(f"Test on mode: {args.task_mode} feat: {group}")
    sort_pos = get_dataset_name(input_columns.keys()) == "rico"
    loss_layer = LossLayer(input_columns)
    total = defaultdict(float)
    nd_columns = model.input_columns
    for step, example in enumerate(dataset):
        if step >= args.steps_per_epoch:
            break
        B, S = example["left"].shape[:2]
        if S == 0:
            continue
        seq_mask = get_seq_mask(example["length"], maxlen=S)
        masks = get_initial_masks(nd_columns, seq_mask)
        if args.task_mode == "random":
            _, masks = random_masking(example, nd_columns, seq_mask, replace_prob=0.0, unchange_prob=0.0)
        elif args.task_mode == "elem":
            eye = torch.eye(S, dtype=torch.bool, device=seq_mask.device)
            example = {k: torch.repeat_interleave(v, S, dim=0) for k, v in example.items()}
            for key, column in nd_columns.items():
                if column["is_sequence"]:
                    masks[key] = eye.repeat(B, 1)
                else:
                    masks[key] = torch.ones(B * S, dtype=torch.bool, device=seq_mask.device)
        else:
            for key in group_keys:
                masks[key] = seq_mask
        demo_args = {"masks": masks, "num_iter": args.num_iter}
        names = get_task_names(input_columns)
        if model.context == "id" and group_name in names:
            demo_args["tasks"] = torch.full(example["left"].shape[:1], names.index(group_name),
                                            device=seq_mask.device)
        prediction = model(example, training=False, demo_args=demo_args)
        if sort_pos and args.task_mode == "pos":
            sort_flag = torch.ones(example["left"].shape[0], dtype=torch.bool, device=seq_mask.device)
            (scores_tmp,) = loss_layer((example, prediction, masks), False, sort_flag)
        else:
            (scores_tmp,) = loss_layer((example, prediction, masks))
        for k, v in scores_tmp.items():
            total[k] += float(v)
    ans = {}
    for k in input_columns:
        num_key, den_key = f"{k}_score_num", f"{k}_score_den"
        if num_key in total.keys():
            ans[k] = total[num_key] / total[den_key] if total[den_key] else float("nan")
    return ans


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--job-dir", required=True, help="The GCS or local path of logs and saved models.")
    parser.add_argument("--batch_size", default=256, type=int)
    parser.add_argument("--task_mode", type=str, default="attr")
    parser.add_argument("--feature", type=str, default="all")
    parser.add_argument("--model", type=str, default="mfp")
    parser.add_argument("--num_iter", type=int, default=1)
    parser.add_argument("--result_csv", type=str, default="")
    args = parser.parse_args(argv)

    with open(os.path.join(args.job_dir, "args.json"), "r") as file_obj:
        train_args = argparse.Namespace(**json.load(file_obj))
    if args.task_mode in ["elem"] and args.batch_size != 1:
        args.batch_size = 1
    logger.info(args)
    device = getattr(train_args, "device", "cuda")
    if device == "cuda":
        device = "cuda:0"
    dataspec = DataSpec(train_args.dataset_name, train_args.data_dir, batch_size=args.batch_size,
                        seq_len=getattr(train_args, "seq_len", None), device=device)
    input_columns = dataspec.make_input_columns()
    dataset = dataspec.make_dataset("test", shuffle=False)
    args.steps_per_epoch = dataspec.steps_per_epoch("test", args.batch_size)
    if args.model != "mfp":
        raise NotImplementedError
    model = MFP(input_columns, latent_dim=train_args.latent_dim, num_blocks=train_args.num_blocks,
                block_type=train_args.block_type, context=train_args.context,
                masking_method=train_args.masking_method, seq_type=train_args.seq_type,
                arch_type=train_args.arch_type, input_dtype=train_args.input_dtype,
                dropout=train_args.dropout, l2=train_args.l2, dtype=getattr(train_args, "dtype", "bf16"),
                device=device)
    weight_path = os.path.join(args.job_dir, "checkpoints", "best.ckpt")
    if not os.path.exists(weight_path + ".safetensors"):
        weight_path = os.path.join(args.job_dir, "checkpoints", "final.ckpt")
    model.compile(optimizer="adam")
    logger.info(f"Loading: {weight_path}")
    model.load_weights(weight_path)
    attribute_groups = get_attribute_groups(input_columns.keys())

    ans_all = {}
    if args.task_mode in ["elem", "random"]:
        ans_all["all"] = evaluate(args, model, dataset, input_columns, None)
    elif args.task_mode == "all_feat":
        for group in attribute_groups.items():
            if group[0] == "type":
                continue
            ans_all[group[0]] = evaluate(args, model, dataset, input_columns, group)
    else:
        group = (args.task_mode, attribute_groups[args.task_mode])
        ans_all[args.task_mode] = evaluate(args, model, dataset, input_columns, group)

    final_results = {}
    for ans in ans_all.values():
        for k, v in ans.items():
            if v == v:
                final_results[k] = round(v, 4)
    print(final_results)
    if args.result_csv:
        with open(args.result_csv, "w") as f:
            writer = csv.writer(f, delimiter=",")
            writer.writerow(list(final_results.keys()))
            writer.writerow(list(final_results.values()))
    return final_results


if __name__ == "__main__":
    main()
