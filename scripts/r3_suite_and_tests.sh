#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mkdir -p gpurun_out/r3
python scripts/suite_sweep.py gpurun_out/r3/suite.json > gpurun_out/r3/suite.md 2> gpurun_out/r3/suite.err || { tail -20 gpurun_out/r3/suite.err; exit 1; }
cat gpurun_out/r3/suite.md
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3/gputest.log 2>&1 || { tail -40 gpurun_out/r3/gputest.log; exit 1; }
tail -3 gpurun_out/r3/gputest.log
