#!/usr/bin/env python3
"""the five real matrices of the suite, 200 products each: for a rocprofv3 kernel trace (which launches a small product is made of, and how long each takes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import sparta_amd as sa
import bench_suite
which = sys.argv[1] if len(sys.argv) > 1 else "ca-HepPh"
for name, kind, make, kw, w in bench_suite.cases(sa):
    if which not in name:
        continue
    r = bench_suite.run_one(sa, torch, name, kind, make, kw, w)
    print(r["name"], r["ms"], r["kernels_ms"], flush=True)
