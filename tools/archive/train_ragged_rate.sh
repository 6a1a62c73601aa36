#!/bin/bash
# End-to-end rate of train.py --samples_per_step on the synthetic MeViS-like RAGGED mix (data drawn by worker processes, text stand-in):
# usage: tools/train_ragged_rate.sh [samples per step = 64] [precision = f16x3] [samples = 1024] [dataset.reader_threads = 16]
set -e
k=${1:-64}; prec=${2:-f16x3}; n=${3:-1024}; nw=${4:-16}
repo=${GRAFT_REPO_ROOT:-/root/repo}
tmp=$(mktemp -d); cd "$tmp"
mkdir -p configs/mevis; sed "s/num_workers: \([0-9]*\)/num_workers: \1\n  reader_threads: $nw/" "$repo/configs/mevis/default.yaml" > configs/mevis/default.yaml
export PYTHONPATH=$repo SOLA_PRECISION=$prec
python "$repo/train.py" --config mevis/default --synthetic true --synthetic_samples $n --synthetic_ragged true --samples_per_step $k --n_epochs_override 2 2>&1 | grep "EPOCH\|Error\|error" | tail -3
