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
# round 4 extras: kernel tables of the power-law configs, the hub kernel's record, the sparse kernels' cache counters, the GPU test log, the lab notes
for src, dst in (("kt_c4_kernel_stats.csv", "c4_kernel_stats.csv"), ("kt_c3_1pct_kernel_stats.csv", "c3_1pct_kernel_stats.csv"), ("kt_c3_0p1pct_kernel_stats.csv", "c3_0p1pct_kernel_stats.csv"),
                 ("hub_profile_5.json", None), ("hub_profile_5_kernel_stats.csv", None), ("hub_profile_1.json", None), ("hub_profile_1_kernel_stats.csv", None),
                 ("sp_pmc_part5.json", "sparse_pmc_c4_part5.json"), ("sp_pmc_part0.json", "sparse_pmc_c4_part0.json"), ("gputest.log", "gpu_tests.log"),
                 ("hub_ubench.txt", "lab_hub_ubench.txt"), ("hub_g4.txt", "lab_hub_g4.txt"), ("hub_ldb.txt", "lab_hub_ldb.txt"), ("hub_ldb2.txt", "lab_hub_ldb2.txt"),
                 ("hub_hot.txt", "lab_hub_hot.txt"), ("hub_lw.txt", "lab_hub_loader_waves.txt"), ("hub_nt.txt", "lab_hub_nt.txt"), ("hub_probe.txt", "lab_hub_probe.txt"),
                 ("hub_big.txt", "lab_hub_big.txt"), ("glds_probe.txt", "lab_glds_probe.txt")):
    s = os.path.join(G, RND, src)
    if os.path.exists(s):
        shutil.copy(s, os.path.join(P, dst or src))
        print("copied", src)
for src, dst in (("r4_hub_parts.txt", "lab_hub_parts.txt"), ("r4_hub_kblock.txt", "lab_hub_kblock.txt"), ("r4_sp_gather_old_f16.txt", "lab_sparse_gather_r3_f16.txt"),
                 ("r4_sp_gather_new_f16.txt", "lab_sparse_gather_scalar_f16.txt"), ("r4_sp_gather_old_bf16.txt", "lab_sparse_gather_r3_bf16.txt"),
                 ("r4_sp_gather_new_bf16.txt", "lab_sparse_gather_scalar_bf16.txt"), ("r4_sp_xcd_c4_0.txt", "lab_sparse_xcd_c4_part0.txt"), ("r4_sp_xcd_c4_5.txt", "lab_sparse_xcd_c4_part5.txt"),
                 ("r4_sp_xcd2_c4_5.txt", "lab_sparse_rowbytes_c4_part5.txt"), ("r4_sp_xcd2_c3_8.txt", "lab_sparse_rowbytes_c3_1pct_part8.txt")):
    s = os.path.join(G, src)
    if RND == "r4" and os.path.exists(s):
        shutil.copy(s, os.path.join(P, dst))
        print("copied", src)
