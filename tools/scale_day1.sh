#!/bin/bash
# scale_day1.sh — first contact with an 8-GPU node (tools/scale_day1.py says what is run and checked); on a one-GPU box:
#   B3W_DIST_BACKEND=gloo tools/scale_day1.sh --gpus-list 1 2 4 --quick --rehearsal
cd "$(dirname "$0")/.." && exec python tools/scale_day1.py "$@"
