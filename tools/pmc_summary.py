"""Per-kernel sums of rocprofv3 --pmc counters (counter_collection.csv) + dispatch counts and durations from the kernel trace
of the same run.  python tools/pmc_summary.py <rocprof output dir> [name filter]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = list(csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    k = k[k.find("::") + 2:][:34] if "anonymous" in k else k[:34]
    if flt and flt not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
dur = collections.defaultdict(float)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if kt:
    for r in csv.DictReader(open(kt[0])):
        k = r["Kernel_Name"]
        k = k[k.find("::") + 2:][:34] if "anonymous" in k else k[:34]
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, c in sorted(agg.items(), key=lambda kv: -dur.get(kv[0], 0)):
    n = len(disp[k])
    line = "%-36s n=%-5d us=%-10.0f" % (k, n, dur.get(k, 0))
    gui = c.get("GRBM_GUI_ACTIVE")
    for name, v in sorted(c.items()):
        line += " %s=%.4g" % (name.replace("SQ_", ""), v)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and gui:
        line += "  | mfma_busy=%.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (gui / 8.0))
    if "SQ_ACTIVE_INST_VALU" in c and gui:
        line += " valu_busy=%.3f" % (c["SQ_ACTIVE_INST_VALU"] * 4 / 1024.0 / (gui / 8.0))
    if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
        w = c["SQ_WAVE_CYCLES"]
        line += " parked=%.3f issue_stall=%.3f active=%.3f" % (c["SQ_WAIT_ANY"] / w, c.get("SQ_WAIT_INST_ANY", 0) / w, c.get("SQ_ACTIVE_INST_ANY", 0) / w)
    print(line)
