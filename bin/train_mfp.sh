#!/usr/bin/env bash
# Drop-in for the reference's bin/train_mfp.sh: same positional/flag surface, PYTHONPATH points
# at this engine's `mfp` package instead of src/mfp.
export PYTHONPATH="$(cd "$(dirname "$0")/.." && pwd)/flex-dm_amd"

DATASET=${1:-"crello"}
NOW=$(date '+%Y%m%d%H%M%S')

DATA_DIR="data/${DATASET}"
[ -d "${DATA_DIR}" ] || DATA_DIR="synthetic"
JOB_DIR="tmp/jobs/${DATASET}/${NOW}"

echo "DATA_DIR=${DATA_DIR}"
echo "JOB_DIR=${JOB_DIR}"

python -m mfp --dataset_name "${DATASET}" --data_dir "${DATA_DIR}" --job-dir "${JOB_DIR}" "${@:2}"
