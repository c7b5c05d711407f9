#!/bin/bash
export WHICH=attnbwd
python tools/bench_fused.py 2>&1 | grep "attn_block_bwd"
