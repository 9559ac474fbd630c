#!/bin/bash
# lab: the sparse-row leg on parts of configs[4] / configs[3]: round-3 gather, scalar gather, XCD-affine streams at several window widths
cfg=${1:-c4}; part=${2:-0}
export HUB_PART=$part
python scripts/lab/r4_hub_parts.py $cfg,only SPARTA_SP_SCALAR=0 SPARTA_SP_XCD=1,SPARTA_SP_WINDOW_COLS=4096 SPARTA_SP_XCD=1,SPARTA_SP_WINDOW_COLS=8192 SPARTA_SP_XCD=1,SPARTA_SP_WINDOW_COLS=16384 SPARTA_SP_XCD=1,SPARTA_SP_WINDOW_COLS=8192,SPARTA_SP_MINSEG=32 SPARTA_SP_XCD=1,SPARTA_SP_WINDOW_COLS=32768 SPARTA_SP_WINDOW_COLS=8192 2>&1 | grep -v Warning
