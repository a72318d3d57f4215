"""Merge a round's traffic_b{1,32}_f16.json (tools/gpu_pmc_ops.sh) into profiles/pmc_traffic.json: the un-suffixed entries that
bench.py attaches as roofline.traffic become the new ones, the previous un-suffixed entries get the suffix given (e.g. _round4).
usage: python tools/r5_merge_traffic.py profiles/r05 _round4"""
import json
import sys

src, suffix = sys.argv[1], sys.argv[2]
path = "profiles/pmc_traffic.json"
d = json.load(open(path))
new = []
for b in (1, 32):
    new += json.load(open(f"{src}/traffic_b{b}_f16.json"))
keys = {(e["key"], e["batch"], e["precision"]) for e in new}
old = []
for e in d["entries"]:
    if (e["key"], e.get("batch"), e.get("precision")) in keys:
        e = dict(e, key=e["key"] + suffix)
    old.append(e)
d["entries"] = new + old
d["note"] += (f"  Merge of {src}: the un-suffixed entries are that round's final build (its tools/gpu_rN_final.sh run); "
              f"*{suffix} = the same launches on the build before it.")
json.dump(d, open(path, "w"), indent=1)
print(len(new), "new entries,", len(old), "kept")
