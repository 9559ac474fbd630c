#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p gpurun_out/r3
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -k "$1" > gpurun_out/r3/newtests.log 2>&1
rc=$?
tail -25 gpurun_out/r3/newtests.log
exit $rc
