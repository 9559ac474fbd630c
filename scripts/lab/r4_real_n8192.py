#!/usr/bin/env python3
"""the reference's real matrices at N = 1024 / 8192 (fixed 64 x 64 grid): sparse-row kernels under SPARTA_SP_ROW_BYTES / other knobs given as NAME=VALUE arguments"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, sparta_amd as sa, bench_suite
variants = [("default", {})] + [(a, dict(x.split("=") for x in a.split(","))) for a in sys.argv[1:]]
for name, kind, make, eng_kw, w in bench_suite.cases(sa, False):
    if not kind.startswith("real"): continue
    m = make()
    for N in (1024, 8192):
        row = []
        for label, env in variants:
            for k, v in env.items(): os.environ[k] = v
            r = bench_suite.run_one(sa, torch, name, kind, None, dict(blocking_algo="fixed_size", row_block_size=64), 64, N=N, budget_ms=100.0, m=m)
            row.append("%s %.4f ms (frac %.3f)" % (label, r["ms"], r["frac_8d"]))
            for k in env: os.environ.pop(k, None)
        print("%-28s N %5d | %s" % (name.split(":")[0], N, " | ".join(row)), flush=True)
