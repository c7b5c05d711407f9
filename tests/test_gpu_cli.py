"""End-to-end plumbing on the GPU through the reference's entry points: ``python -m mfp`` (config
c1 of BASELINE.json: RICO, masking_method=random, 2 blocks, d_model=128, seq_len=32, batch=8) and
``eval.py`` on the job directory it wrote."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def test_train_cli_rico_then_eval(tmp_path, capsys):
    from mfp.main import main
    job = str(tmp_path / "job")
    main(["--dataset_name", "rico", "--data_dir", "synthetic:32:32", "--job-dir", job, "--latent_dim", "128",
          "--num_blocks", "2", "--batch_size", "8", "--num_epochs", "2", "--validation_freq", "1",
          "--masking_method", "random", "--dtype", "fp32", "--verbose", "0"])
    out = capsys.readouterr().out
    assert "total_score" in out and "loss" in out                      # train.py:90-92 prints metric lines
    args = json.load(open(os.path.join(job, "args.json")))             # train.py:30-33
    assert args["dataset_name"] == "rico" and args["latent_dim"] == 128 and args["job_dir"] == job
    ck = os.path.join(job, "checkpoints")
    assert os.path.exists(os.path.join(ck, "final.ckpt.safetensors"))  # train.py:95-97
    assert os.path.exists(os.path.join(ck, "best.ckpt.safetensors"))   # callbacks.py:49-56
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("eval_cli", os.path.join(root, "eval.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    res = ev.main(["--job-dir", job, "--task_mode", "attr", "--batch_size", "8"])
    assert set(res) == {"icon", "clickable", "text_button"} and all(0.0 <= v <= 1.0 for v in res.values())
    res = ev.main(["--job-dir", job, "--task_mode", "random", "--batch_size", "8", "--num_iter", "2"])
    assert "left" in res
    res = ev.main(["--job-dir", job, "--task_mode", "elem"])
    assert "type" in res


def test_train_cli_crello_bf16_graph(tmp_path):
    from mfp.main import main
    job = str(tmp_path / "job2")
    main(["--dataset_name", "crello", "--data_dir", "synthetic:16:32", "--job-dir", job, "--latent_dim", "128",
          "--num_blocks", "1", "--batch_size", "16", "--num_epochs", "2", "--validation_freq", "2",
          "--masking_method", "elem_pos_attr_img_txt", "--dtype", "bf16", "--use_graph", "--verbose", "0",
          "--enable_profile", "--context", "id"])
    assert os.path.exists(os.path.join(job, "checkpoints", "final.ckpt.safetensors"))
    # --enable_profile = TensorBoard(profile_batch=2) of the reference (callbacks.py:44-48): the 2nd
    # train step is traced; its kernel table must name kernels of the HIP library
    table = os.path.join(job, "logs", "profile_step2.kernels.txt")
    assert os.path.exists(table) and os.path.exists(os.path.join(job, "logs", "profile_step2.trace.json"))
    txt = open(table).read()
    assert "gemm" in txt or "attn" in txt, txt[:2000]
    # eval.py on a context="id" job sends the task id of the evaluated group (eval.py:96-100)
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("eval_cli2", os.path.join(root, "eval.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    res = ev.main(["--job-dir", job, "--task_mode", "pos", "--batch_size", "16"])
    assert set(res) == {"left", "top", "width", "height"}


def test_train_cli_on_tfrecord_directory(tmp_path, capsys):
    """--data_dir pointing at <split>-*.tfrecord files (the reference's dataset layout): read by the
    TensorFlow-free reader, variable sequence length per batch, eager f32 steps."""
    from mfp.data.spec import write_synthetic_tfrecords
    from mfp.main import main
    data = str(tmp_path / "crello")
    # 27 train documents at batch 8: batches straddle the epoch boundary (shuffle -> repeat -> batch,
    # spec.py:244-248) and are always full; the 11 val/test documents end with a short batch
    write_synthetic_tfrecords(data, "crello", {"train": 27, "val": 11, "test": 11}, seq_len=9, seed=2)
    job = str(tmp_path / "job3")
    main(["--dataset_name", "crello", "--data_dir", data, "--job-dir", job, "--latent_dim", "128",
          "--num_blocks", "1", "--batch_size", "8", "--num_epochs", "2", "--validation_freq", "1",
          "--masking_method", "random", "--dtype", "fp32", "--verbose", "0"])
    out = capsys.readouterr().out
    assert "total_score" in out
    assert os.path.exists(os.path.join(job, "checkpoints", "final.ckpt.safetensors"))


def test_default_command_line_runs_the_document_tile_kernels(tmp_path, capsys):
    """bin/train_mfp.sh as the reference documents it -- no --seq_len -- on a TFRecord directory (bin/train_mfp.sh:16-20,
    src/mfp/mfp/data/spec.py:255-276): the batches are padded to 64 positions and stepped on the document-tile kernels
    (attn_block_fwd_kernel with SDOC = 64: one launch per block forward), and ``--seq_len 0`` keeps the reference's per-batch
    lengths on the generic kernels."""
    from mfp.data.spec import write_synthetic_tfrecords
    from mfp.main import main
    data = str(tmp_path / "crello")
    write_synthetic_tfrecords(data, "crello", {"train": 32, "val": 8, "test": 8}, seq_len=11, seed=5)

    def run(job, *extra):
        main(["--dataset_name", "crello", "--data_dir", data, "--job-dir", job, "--num_blocks", "1", "--batch_size", "8",
              "--num_epochs", "1", "--validation_freq", "1", "--verbose", "0", "--enable_profile", *extra])
        out = capsys.readouterr().out
        assert "total_score" in out and os.path.exists(os.path.join(job, "checkpoints", "final.ckpt.safetensors"))
        return out, open(os.path.join(job, "logs", "profile_step2.kernels.txt")).read()

    out, kernels = run(str(tmp_path / "job_default"))      # latent_dim 256, bf16: the flags' defaults
    assert "attn_block_fwd_kernel" in kernels and "mfp.train:" not in out, kernels[:3000]
    import re
    targs = [m.group(1).replace(" ", "").split(",") for m in re.finditer(r"attn_block_fwd_kernel<([^>]*)>", kernels)]
    assert targs and all(len(a) >= 4 and a[3] == "64" for a in targs), targs      # <DROPOUT, MLP, STASH, SDOC, ...>
    out, kernels = run(str(tmp_path / "job_ragged"), "--seq_len", "0")
    assert "attn_block_fwd_kernel" not in kernels and "falls off the document-tile kernels" in out


def test_weights_from_tensorflow_checkpoint(tmp_path, capsys):
    """--weights / load_weights on a TensorFlow checkpoint prefix (train.py:67-69): the bundle is
    read without TensorFlow and every variable lands where the Keras object graph says."""
    import numpy as np
    import torch
    from mfp.data import tf_checkpoint as tfc
    from mfp.data.spec import make_input_columns, synthetic_batch
    from mfp.main import main
    from mfp.models.mfp import MFP
    ic = make_input_columns("rico")
    src = MFP(ic, num_blocks=2, latent_dim=128, dtype="fp32", device="cuda:0", seed=5)
    state = src.model.store.state_dict()
    prefix = str(tmp_path / "pretrained" / "best.ckpt")
    tfc.write_bundle(prefix, {tfc.checkpoint_key(k): v.numpy() for k, v in state.items()}, snappy=True)
    dst = MFP(ic, num_blocks=2, latent_dim=128, dtype="bf16", device="cuda:0", seed=6)
    assert not torch.equal(dst.model.store.state_dict()["decoder/decoder_type/kernel"], state["decoder/decoder_type/kernel"])
    dst.load_weights(prefix)
    for k, v in dst.model.store.state_dict().items():
        assert torch.equal(v, state[k]), k
    # same predictions from both models (the bf16 shadow weights were refreshed by the load)
    batch = synthetic_batch(ic, 4, 16, seed=1, ragged=True, device="cuda:0")
    dst32 = MFP(ic, num_blocks=2, latent_dim=128, dtype="fp32", device="cuda:0", seed=7).load_weights(prefix)
    a = src.model(batch, training=False)
    b = dst32.model(batch, training=False)
    for k in ("type", "left", "icon"):
        assert torch.equal(a[k], b[k]), k
    wrong = MFP(ic, num_blocks=1, latent_dim=128, dtype="fp32", device="cuda:0")
    with pytest.raises(ValueError, match="unmatched"):
        wrong.load_weights(prefix)
    job = str(tmp_path / "job4")
    main(["--dataset_name", "rico", "--data_dir", "synthetic:16:16", "--job-dir", job, "--latent_dim", "128",
          "--num_blocks", "2", "--batch_size", "8", "--num_epochs", "1", "--validation_freq", "1",
          "--masking_method", "random", "--dtype", "fp32", "--verbose", "0", "--weights", prefix])
    assert "total_score" in capsys.readouterr().out


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """bench.py as the driver launches it for N > 1 (``python -m torch.distributed.run ... bench.py --gpus N``), here with
    two ranks sharing the one GPU over gloo (MFP_DIST_BACKEND): the line must come out -- every rank takes part in every
    collective of the script, the roofline leg included (rank 0 alone stepping eagerly would wait for its peers forever) --
    with the replicas in sync and the all-reduce plan on it."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MFP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--config", "c4", "--batch", "16"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["params_in_sync"] is True and line["value"] > 0
    assert line["dp"]["collective_ranks"] == 2 and sum(b["params"] for b in line["dp"]["plan"]) == line["config"]["params"]
    assert "roofline" in line and "cpu_baseline" not in line
