#!/usr/bin/env python
"""Experiment: does running two half-batch chains of weight-stationary GEMMs on two streams (each sized
for half the CUs: MFP_WS_NCU=128) fill the prologue / tail bubbles of one full-size chain?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flex-dm_amd")]
import torch
from mfp.hip import ops
D = 256
halves = int(os.environ.get("HALVES", 1))
T = 32768 // halves
dev = "cuda"
def mk():
    y = torch.randn(T, D, device=dev).to(torch.bfloat16)
    Wqkv = torch.randn(3 * D, D, device=dev).to(torch.bfloat16); bq = torch.zeros(3 * D, device=dev)
    W1 = torch.randn(2 * D, D, device=dev).to(torch.bfloat16); b1 = torch.zeros(2 * D, device=dev)
    W2 = torch.randn(D, 2 * D, device=dev).to(torch.bfloat16); b2 = torch.zeros(D, device=dev)
    x = torch.randn(T, D, device=dev)
    return y, Wqkv, bq, W1, b1, W2, b2, x
def chain(a):
    y, Wqkv, bq, W1, b1, W2, b2, x = a
    for _ in range(4):
        ops.gemm(y, Wqkv, T, 3 * D, D, a_kmajor=True, b_kmajor=True, bias=bq, out_dtype=torch.bfloat16)
        h = ops.gemm(y, W1, T, 2 * D, D, a_kmajor=True, b_kmajor=True, bias=b1, relu=True, out_dtype=torch.bfloat16)
        ops.gemm(h, W2, T, D, 2 * D, a_kmajor=True, b_kmajor=True, bias=b2, residual=x, out_dtype=torch.float32)
args = [mk() for _ in range(halves)]
streams = [torch.cuda.Stream() for _ in range(halves)]
def run():
    for a, s in zip(args, streams):
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            chain(a)
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
for _ in range(3): run()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): g.replay()
e1.record(); torch.cuda.synchronize()
print("halves=%d ncu=%s: %.1f us per 12-GEMM chain set" % (halves, os.environ.get("MFP_WS_NCU"), 1e3 * e0.elapsed_time(e1) / 20))
