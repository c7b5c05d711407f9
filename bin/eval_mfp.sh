#!/usr/bin/env bash
# Drop-in for the reference's bin/eval_mfp.sh
python3 "$(cd "$(dirname "$0")/.." && pwd)/eval.py" "$@"
