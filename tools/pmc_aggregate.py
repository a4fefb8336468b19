"""Fold the counter_collection CSVs of tools/collect_pmc.sh into one JSON:
{kernel: {counter: {"launches": n, "mean_per_launch": v}}} (FETCH_SIZE / WRITE_SIZE in KiB as
rocprofv3 reports them; the gfx950 x2 correction of FETCH_SIZE is applied by the reader)."""
import collections, csv, glob, json, re, sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void \(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"])
        per[(name, int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"])
    for (name, _, ctr), v in per.items():
        acc[name][ctr].append(v)
out = {}
for name, ctrs in acc.items():
    out[name] = {}
    for ctr, vals in ctrs.items():
        key = "mean_per_launch_KiB" if ctr in ("FETCH_SIZE", "WRITE_SIZE") else "mean_per_launch"
        out[name][ctr] = {"launches": len(vals), key: sum(vals) / len(vals)}
print(json.dumps(out, indent=1, sort_keys=True))
