"""Per-kernel averages of a rocprofv3 --pmc --kernel-trace csv pair:  python tools/pmc_report.py <dir> [name-filter]"""
import collections, csv, glob, sys
d = sys.argv[1]; needle = sys.argv[2] if len(sys.argv) > 2 else "lslam"
f = glob.glob(d + "/*/*_counter_collection.csv")[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f.replace("counter_collection", "kernel_trace"))):
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in per.items():
    if needle not in k: continue
    print(k[:90], f"launches {len(dur[k])}  avg {sum(dur[k])/len(dur[k]):.1f} us  max {max(dur[k]):.1f} us")
    for c, x in sorted(v.items()):
        print(f"    {c:28s} avg {sum(x)/len(x):16.1f}   max {max(x):16.1f}")
