"""lab: a readable summary of a bench.py line (python scripts/lab/r5_bench_summary.py file.json)"""
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: j[k] for k in ("metric", "value", "ms_per_step", "steps")}, "roofline frac", j["roofline"]["frac"], "achieved", j["roofline"]["achieved"])
s = j["config"].get("suite")
if s:
    for r in s["matrices"]:
        print("%-50s ms %.4f frac %.3f %s" % (r["name"][:50], r.get("ms", -1), r.get("frac_8d", -1), r.get("carried_by", r.get("error", ""))[:100]))
    print({k: s[k] for k in s if k.startswith("median") or k.startswith("real_median") or k in ("seconds", "skipped_for_time", "bound_violations", "check_failures", "min_frac_8d")})
    for r in s.get("real_matrix_sweep", []):
        print(r["name"][:34], [(p.get("n_cols"), p.get("arm", "")[:6], p.get("ms"), p.get("frac_8d")) for p in r["points"]])
    for r in s.get("block_size_sweep", []):
        print(r["name"][:34], [(p.get("block"), p.get("arm", "")[:8], p.get("n_cols"), p.get("ms"), p.get("frac_8d"), p.get("error")) for p in r.get("points", [])], r.get("error"))
