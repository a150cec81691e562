"""Summarise rocprofv3 PMC csv output: mean per launch of every counter for kernels matching a name.
usage: python profiles/summarize_pmc.py <dir with *counter_collection.csv> <kernel substring> <out.json> [bench steps incl. warm-up]
With the step count the summary also says how many launches of the kernel one bench step makes (`launches_per_step`: the
chunked traceback pipelines launch their sweep several times per step), so that bench.py can quote HBM traffic per step."""
import csv, glob, json, os, sys
from collections import defaultdict


def main(root, needle, out, steps=0):
    steps = int(steps)
    acc = defaultdict(lambda: defaultdict(float))       # counter -> dispatch id -> sum over dimensions
    kname = None
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if needle not in row["Kernel_Name"]:
                    continue
                acc[row["Counter_Name"]][(path, row["Dispatch_Id"])] += float(row["Counter_Value"])
    res = {c: {"launches": len(d), "mean_per_launch": sum(d.values()) / len(d)} for c, d in acc.items()}
    res["kernel"] = needle
    if steps:
        n = max(v["launches"] for v in res.values() if isinstance(v, dict))
        res["launches_per_step"] = n / float(steps)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({c: round(v["mean_per_launch"]) for c, v in res.items() if isinstance(v, dict)}))


if __name__ == "__main__":
    main(*sys.argv[1:5])
