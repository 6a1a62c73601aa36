#!/bin/bash
# End-to-end rate of train.py on synthetic uniform-shape data at a batch size.  usage: tools/train_rate.sh [batch] [precision] [samples]
set -e
b=${1:-64}; prec=${2:-f16x3}; n=${3:-1024}
repo=${GRAFT_REPO_ROOT:-/root/repo}
tmp=$(mktemp -d); cd "$tmp"
mkdir -p configs/mevis; sed "0,/batch_size: 1/s//batch_size: $b/" "$repo/configs/mevis/default.yaml" > configs/mevis/default.yaml
export PYTHONPATH=$repo SOLA_PRECISION=$prec
python "$repo/train.py" --config mevis/default --synthetic true --synthetic_samples $n --synthetic_tracks 64 --synthetic_frames 32 --n_epochs_override 2 2>&1 | grep EPOCH
