"""Summarise a rocprofv3 --pmc counter_collection.csv for one kernel name substring."""
import collections, csv, glob, sys
d = sys.argv[1]; pat = sys.argv[2]; items = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
f = glob.glob(d + "/*/*counter_collection.csv")[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if pat not in r["Kernel_Name"]:
        continue
    k = int(r["Dispatch_Id"])
    per.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    per[k]["dur_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, v in per.items():
    print(k, " ".join(f"{c}={x:.4g}" if c == "dur_ms" else f"{c}/item={x/items:.1f}" for c, x in v.items()))
