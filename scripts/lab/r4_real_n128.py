#!/usr/bin/env python3
"""the reference's real matrices at N = 128 (suite blocking): product time under the current environment"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, sparta_amd as sa, bench_suite
for name, kind, make, eng_kw, w in bench_suite.cases(sa, False):
    if not kind.startswith("real"): continue
    for N in (128, 512):
        r = bench_suite.run_one(sa, torch, name, kind, make, eng_kw, w, N=N, budget_ms=100.0)
        print("%-28s N %4d | %.4f ms (prepared B %s) frac %.3f kernels %s check %.1e" % (name.split(":")[0], N, r["ms"], r.get("ms_prepared_b"), r["frac_8d"], r["kernels_ms"], r["check_max_err"]), flush=True)
