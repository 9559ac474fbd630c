#!/usr/bin/env python3
"""gpurun_out/ (what scripts/profile_bench.sh + scripts/r4_parts_full.sh left) -> profiles/r3/ + profiles/traffic*.json, named by kernel revision.
    python scripts/collect_profiles.py r4 r4a"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND, rev = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles", RND)


def last_json_line(path):
    return [l for l in open(path) if l.startswith("{")][-1]


for tag, name, traffic in ((RND + "_default", "bench_default", "traffic.json"), (RND + "_f16", "bench_f16", "traffic_f16.json")):
    src = os.path.join(G, "prof_" + tag)
    if not os.path.isdir(src):
        print("no", src)
        continue
    for f in glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.reader(open(f)))
        with open(os.path.join(P, "%s_%s_kernel_stats.csv" % (name, rev)), "w", newline="") as o:
            csv.writer(o, quoting=csv.QUOTE_NONNUMERIC).writerows([rows[0]] + [[r[0]] + [float(x) if "." in x or "e" in x else int(x) for x in r[1:]] for r in rows[1:6]])
    for pm in ("fetch", "write", "sq"):
        agg = collections.defaultdict(list)
        for f in glob.glob(src + "/pmc_%s/**/*counter_collection.csv" % pm, recursive=True):
            for r in csv.DictReader(open(f)):
                agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        with open(os.path.join(P, "%s_%s_pmc_%s.csv" % (name, rev, pm)), "w", newline="") as o:
            wr = csv.writer(o)
            wr.writerow(["Kernel_Name", "Counter_Name", "launches", "mean", "min", "max"])
            for (k, c), v in agg.items():
                wr.writerow([k, c, len(v), sum(v) / len(v), min(v), max(v)])
    shutil.copy(os.path.join(src, "summary.json"), os.path.join(P, "%s_%s_summary.json" % (name, rev)))
    open(os.path.join(P, "%s_%s_line_under_rocprof.json" % (name, rev)), "w").write(last_json_line(os.path.join(src, "bench_trace.log")))
    if os.path.exists(os.path.join(src, "traffic.json")):
        shutil.copy(os.path.join(src, "traffic.json"), os.path.join(ROOT, "profiles", traffic))
for src, dst in (("bench_driver_cmd.json", "bench_driver_cmd_line.json"), ("bench_default.json", "bench_default_line.json"), ("bench_f16.json", "bench_f16_line.json"),
                 ("slabs8_scale23.json", None), ("c3_0p1pct_off.json", None), ("c3_0p1pct_on.json", None), ("c3_0p1pct_auto.json", None), ("c3_1pct_off.json", None),
                 ("c3_5pct_off.json", None), ("half_of_c4.json", None), ("suite.json", None), ("suite.md", None)):
    s = os.path.join(G, RND, src)
    if not os.path.exists(s):
        print("no", s)
        continue
    d = os.path.join(P, dst or src)
    if src.endswith(".json") and src != "suite.json":
        open(d, "w").write(last_json_line(s))
    else:
        shutil.copy(s, d)
    print("copied", src)
