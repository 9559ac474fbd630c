"""Developer script: the benchmark set of bench_suite.py (synthetic families + the reference's real matrices + the 20 M-nonzero R-MAT) on one MI355X,
fp32, N = 128 -- the same records bench.py appends to its line as config.suite, as a markdown table on stdout and a JSON file.
    python scripts/suite_sweep.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sparta_amd as sa  # noqa: E402
import bench_suite  # noqa: E402

OUT = sys.argv[1] if len(sys.argv) > 1 else None
print("| matrix | rows | nnz | blocking (host s) | ms | useful GFLOP/s | carried by | frac_8d | gather GB/s | ms behind a prepared B |")
print("|---|---|---|---|---|---|---|---|---|---|")


def log(r):
    if "points" in r and any("block" in p_ for p_ in r["points"]):      # the reference's block sizes: a real matrix under the fixed grid at 128 .. 1024
        for p_ in r["points"]:
            print("|   %s, %d x %d blocks, N = %d, fixed grid | | | | %s | %s | %s | %s | |" % (r["name"].split(":")[0], p_["block"], p_["block"], p_["n_cols"],
                  p_.get("ms", p_.get("error")), p_.get("useful_gflops", ""), p_.get("carried_by", ""), p_.get("frac_8d", "")), flush=True)
        return
    if "points" in r and "reference_csv" in r:                # the R-MAT of the reference's published row, 1024 x 1024 blocks, N = 8192
        for p_ in r["points"]:
            print("|   %s, %s (%.2f s) | %d | %d | | %.4f | %.0f | %s | %.3f | reference (its GPU, context only): %.2f / %.2f ms |" % (r["name"], p_["arm"], p_["host_reorder_s"], r["rows"], r["nnz"],
                  p_["ms"], p_["useful_gflops"], p_["carried_by"], p_["frac_8d"], r["reference_csv"]["fixed_grid_ms"], r["reference_csv"]["clustered_ms"]), flush=True)
        return
    if "points" in r:                                        # a real matrix at the reference's operand widths, its two arms + blocking_algo 7
        for p_ in r["points"]:
            if "error" in p_:
                print("|   %s N = %d, %s | error: %s |" % (r["name"].split(":")[0], p_["n_cols"], p_["arm"], p_["error"]), flush=True)
            else:
                print("|   %s N = %d, %s (%.2f s) | | | | %.4f | %.0f | %s | %.3f | check %.1e |" % (r["name"].split(":")[0], p_["n_cols"], p_["arm"], p_["host_reorder_s"], p_["ms"],
                      p_["useful_gflops"], p_["carried_by"], p_["frac_8d"], p_["check_max_err"]), flush=True)
        return
    if "error" in r:
        print("| %s | error: %s |" % (r["name"], r["error"]), flush=True)
        return
    print("| %s | %d | %d | %s (%.2f) | %.4f | %.0f | %s | %.3f | %s | %s |" % (r["name"], r["rows"], r["nnz"], r["blocking"], r["host_seconds"]["reorder"], r["ms"],
          r["useful_gflops"], r["carried_by"], r["frac_8d"], "-" if r["gather_gbs"] is None else "%.0f" % r["gather_gbs"],
          "-" if r.get("ms_prepared_b") is None else "%.4f" % r["ms_prepared_b"]), flush=True)


res = bench_suite.run(sa, torch, N=128, device=0, large=True, time_budget_s=600.0, log=log, sweep_budget_s=600.0, block_budget_s=600.0)
res["device"] = torch.cuda.get_device_name(0)
res["kernel_rev"] = sa.KERNEL_REV
print("min frac_8d %.3f, median %.3f, %.1f s" % (res["min_frac_8d"], res["median_frac_8d"], res["seconds"]))
print({k: v for k, v in res.items() if k.startswith("real_median") or k in ("bound_violations", "check_failures", "skipped_for_time")})
if OUT:
    json.dump(res, open(OUT, "w"), indent=1)
