"""The block-shaped grouped weight-gradient launch, a few times: target of tools/pmc_kernel.sh."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd"), os.path.join(ROOT, "tools")]
import torch
from mfp.hip import ops
import bench_wgrad
jobs = bench_wgrad.block_jobs()
for _ in range(10):
    ops.wgrad_group(jobs, bench_wgrad.T, int(os.environ.get("SK", 8)))
torch.cuda.synchronize()
