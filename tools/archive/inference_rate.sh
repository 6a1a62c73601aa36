#!/bin/bash
# End-to-end rate of inference.py on the synthetic MeViS-like ragged mix (text stand-in, PNG-free): train one tiny epoch for a
# checkpoint, then score N samples with up to 128 samples per ragged call.  usage: tools/inference_rate.sh [samples] [precision]
set -e
n=${1:-1024}; prec=${2:-f16x3}
repo=${GRAFT_REPO_ROOT:-/root/repo}
tmp=$(mktemp -d); cd "$tmp"
mkdir -p configs/mevis; cp "$repo/configs/mevis/default.yaml" configs/mevis/default.yaml
export PYTHONPATH=$repo SOLA_PRECISION=$prec
python "$repo/train.py" --config mevis/default --synthetic true --synthetic_samples 4 --synthetic_tracks 8 --synthetic_frames 16 --n_epochs_override 1 > train.log 2>&1
for mx in 1 128; do
  python "$repo/inference.py" --config mevis/default --synthetic true --synthetic_samples $n --synthetic_ragged true --eval_weight_epoch 1 --ragged_max_samples $mx 2>&1 | grep "samples/s" | sed "s/^/max $mx per call: /"
done
