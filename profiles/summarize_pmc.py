"""Summarise rocprofv3 PMC csv output: mean per launch of every counter for kernels matching a name.
usage: python profiles/summarize_pmc.py <dir with *counter_collection.csv> <kernel substring> <out.json> [bench steps incl. warm-up]
With the step count the summary also says how many launches of the kernel one bench step makes (`launches_per_step`: the
chunked traceback pipelines launch their sweep several times per step), so that bench.py can quote HBM traffic per step.
The summary records the git blob id of the kernel's source file (`source_blobs`): bench.py quotes counters only while the file
is unchanged (a kernel edit that keeps the kernel's name must not report stale counters)."""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parasail-rs_amd", "csrc")


def git_blob_id(path):
    """what `git rev-parse HEAD:<path>` prints for a committed, unmodified file (sha1 of "blob <size>\0" + content)"""
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def kernel_source(kernel):
    """csrc file of a kernel name: pmx_nwsg16q_kernel<...> -> pmx_nwsg16.hip (the longest file stem that prefixes the name)"""
    m = re.search(r"pmx_[a-z0-9_]+", kernel)
    if not m:
        return None
    name = m.group(0)
    best = None
    for f in glob.glob(os.path.join(CSRC, "pmx_*.hip")):
        stem = os.path.basename(f)[:-4]
        if name.startswith(stem) and (best is None or len(stem) > len(os.path.basename(best)) - 4):
            best = f
    return best


def source_blobs(kernel):
    src = kernel_source(kernel)
    # the dispatcher and the shared header too: an edit to the chunking changes launches_per_step under the same kernel name
    files = [f for f in (src, os.path.join(CSRC, "pmx_api.hip"), os.path.join(CSRC, "pmx_common.h")) if f and os.path.exists(f)]
    return {os.path.basename(f): git_blob_id(f) for f in files}


def main(root, needle, out, steps=0):
    steps = int(steps)
    acc = defaultdict(lambda: defaultdict(float))       # counter -> dispatch id -> sum over dimensions
    names = set()
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if needle not in row["Kernel_Name"]:
                    continue
                names.add(row["Kernel_Name"].split("(")[0])
                acc[row["Counter_Name"]][(path, row["Dispatch_Id"])] += float(row["Counter_Value"])
    res = {c: {"launches": len(d), "mean_per_launch": sum(d.values()) / len(d)} for c, d in acc.items()}
    res["kernel"] = needle
    res["kernel_names"] = sorted(names)
    res["source_blobs"] = source_blobs(needle)
    if steps and acc:
        n = max(v["launches"] for v in res.values() if isinstance(v, dict) and "launches" in v)
        res["launches_per_step"] = n / float(steps)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({c: round(v["mean_per_launch"]) for c, v in res.items() if isinstance(v, dict) and "mean_per_launch" in v}))


if __name__ == "__main__":
    main(*sys.argv[1:5])
