"""HBM traffic per launch of the image gather kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
tools/image_gather_probe.py: per (kernel, grid size) the mean counters and the bytes after the gfx950 correction of
MI355X_MICROARCH.md (bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024), next to nothing else - the algorithmic bytes are in the
probe's own JSON.

    python tools/pmc_image_gather.py <fetch pass dir> <write pass dir>
"""
import collections
import csv
import glob
import sys


def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("image_" in r["Kernel_Name"]):
                acc[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Grid_Size", "?"))].append(float(r["Counter_Value"]))
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
for key in sorted(set(fetch) | set(write)):
    f, w = fetch.get(key, [0.0]), write.get(key, [0.0])
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    print(f"{key[0]:60s} grid {key[1]:>9s}  launches {len(f):3d}  FETCH_SIZE KiB avg {fm:10.0f}  WRITE_SIZE KiB avg {wm:10.0f}  "
          f"HBM bytes (2 x fetch + write) {(2 * fm + wm) * 1024 / 1e6:8.1f} MB")
