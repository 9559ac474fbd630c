#!/bin/bash
# lab: bytes of a row of B per gather instruction x window width, XCD-affine streams
cfg=${1:-c4}; part=${2:-0}
export HUB_PART=$part
python scripts/lab/r4_hub_parts.py $cfg,only SPARTA_SP_XCD=1,SPARTA_SP_WINDOW_COLS=8192 SPARTA_SP_XCD=1,SPARTA_SP_WINDOW_COLS=8192,SPARTA_SP_ROW_BYTES=512 SPARTA_SP_XCD=1,SPARTA_SP_WINDOW_COLS=4096,SPARTA_SP_ROW_BYTES=512 SPARTA_SP_XCD=1,SPARTA_SP_WINDOW_COLS=4096,SPARTA_SP_ROW_BYTES=1024 SPARTA_SP_XCD=1,SPARTA_SP_WINDOW_COLS=2048,SPARTA_SP_ROW_BYTES=1024 2>&1 | grep -v "Warning\|amdgpu.ids" | cut -c1-140
