import sys, glob, csv, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "vbs_spmm" not in n: continue
        mt = re.search(r"vbs_spmm\w*<[^>]*>|vbs_spmm\w*", n)
        k = mt.group(0) + " vgpr=%s agpr=%s lds=%s" % (r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = []
for k, d in agg.items():
    lines.append(k + "  (profiled dispatch mean %.1f us)" % (sum(dur[k]) / len(dur[k]) / 1e3))
    for c, v in sorted(d.items()):
        lines.append("   %-28s n=%d mean=%.5g" % (c, len(v), sum(v) / len(v)))
txt = "\n".join(lines)
open(out + "/summary.txt", "w").write(txt + "\n")
print(txt)
