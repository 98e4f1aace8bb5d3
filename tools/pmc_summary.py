#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, mean counter value per dispatch
and mean duration.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide
coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -- both raw and corrected are printed."""
import csv, glob, sys, collections, re

def short(name):
    m = re.search(r"(k_[a-z_0-9]+)<([^>]*)>", name)
    if m: return "%s<%s>" % (m.group(1), m.group(2))
    if "radix_sort" in name or "merge_sort" in name: return "rocprim_sort"
    return name[:40]

def main(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            agg[k]["_vgpr"].append(float(r["VGPR_Count"])); agg[k]["_lds"].append(float(r["LDS_Block_Size"]))
    for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1]["_dur_us"])):
        if not k.startswith("k_") and "sort" not in k: continue
        n = len(c["_dur_us"])
        line = "%-34s n=%-4d dur_us=%9.1f vgpr=%3d lds=%6d" % (k, n, sum(c["_dur_us"]) / n, c["_vgpr"][0], c["_lds"][0])
        for name, v in sorted(c.items()):
            if name.startswith("_"): continue
            m = sum(v) / len(v)
            if name == "FETCH_SIZE":
                line += "  FETCH=%.1f MiB (x2 corrected %.1f MiB)" % (m / 1024, 2 * m / 1024)
            elif name == "WRITE_SIZE":
                line += "  WRITE=%.1f MiB" % (m / 1024)
            else:
                line += "  %s=%.4g" % (name, m)
        print(line)

if __name__ == "__main__":
    for d in sys.argv[1:]:
        print("==", d); main(d)
