"""Developer script: rebuild profiles/r2/configs.json from the bench lines scripts/r2_collect.sh and scripts/r2_configs_big.sh left under gpurun_out/r2/
(the config descriptions are kept from the existing file)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(ROOT, "profiles", "r2", "configs.json")
old = json.load(open(path))
lines = []
for e in old["lines"]:
    f = os.path.join(ROOT, "gpurun_out", "r2", e["file"] + ".json")
    try:
        line = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as ex:
        print("kept the old line of", e["file"], "(", ex, ")")
        line = e["line"]
    lines.append({"config": e["config"], "file": e["file"], "line": line})
note = ("one bench.py JSON line per BASELINE config / arm, MI355X (one GPU box per gpurun call: the c1 / c2 lines come from one box, the others from another), end of "
        "round 2 (kernel revision r2c: no-barrier one-tile kernels, longest-tile-first dealing, per-block tile / sparse-row split of sparta_vbs_create_from_csr, "
        "TAIL-free instantiations, 16-bit slices of A in step order).  Box to box the same binary spreads: fp32 flagship 48.7-49.8 us at 2000 steps, ogbn-like "
        "10.3-11.7 ms (two groups of boxes, ~10.5 and ~11.6).  Earlier in the round: r2b flagship 49.3-49.9 us, f16 22.8-23.0 us; r2a driver command 59.8 us / "
        "0.626, f16 36.5 us / 0.318, ogbn-like 11.5-13.2 ms, 0.012 % 21.5 / 21.5 ms, 0.1 % 182 / 201 ms.")
json.dump({"note": note, "commands": old["commands"], "lines": lines}, open(path, "w"), indent=1)
for l in lines:
    L = l["line"]
    print("%-34s %10s ms  value %10s  frac %s" % (l["file"], L.get("ms_per_step"), L.get("value"), (L.get("roofline") or {}).get("frac")))
