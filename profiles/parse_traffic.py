"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) into
profiles/traffic.json: average HBM bytes per launch for each of our kernels, corrected as
MI355X_MICROARCH.md prescribes for gfx950: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE under-reports wide coalesced reads by exactly 2x; both counters are in KiB).

    python profiles/parse_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [<fetch2> <write2> ...]

Further pairs (the --amp run's passes) are merged into the same table: their kernels have names of their own.  rocprofv3
leaves names with _Float16 parameters mangled (its demangler does not know DF16_); `demangle` below recovers the kernel
name and its template arguments from the Itanium form, which is all the table keys need.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def demangle(name):
    """_Z<len><name>I<template args>E<parameters>  ->  name<args>  (integral / bool literals and a few type names)."""
    m = re.match(r"_Z(\d+)", name)
    if not m:
        return name
    n, pos = int(m.group(1)), m.end()
    base, rest = name[pos:pos + n], name[pos + n:]
    if not rest.startswith("I"):
        return base
    args, i = [], 1
    types = {"f": "float", "d": "double", "i": "int", "DF16_": "_Float16", "h": "unsigned char", "l": "long"}
    while i < len(rest) and rest[i] != "E":
        if rest[i] == "L":                                   # literal: L <type letter> <digits> E
            j = rest.index("E", i)
            ty, val = rest[i + 1], rest[i + 2:j]
            val = val.replace("n", "-")
            args.append(("true" if val == "1" else "false") if ty == "b" else val)
            i = j + 1
            continue
        for code, text in types.items():
            if rest.startswith(code, i):
                args.append(text)
                i += len(code)
                break
        else:
            return base                                      # something this parser does not know: name only
    return base + "<" + ",".join(args) + ">"


def short(name):
    if name.startswith("_Z"):
        name = demangle(name)
    m = re.match(r"(?:void )?([A-Za-z_0-9]+)(<[^>]*>)?", name)
    if not m:
        return name
    base, targs = m.group(1), (m.group(2) or "")
    return base + targs.replace(" ", "")


def collect(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            tot[k] += float(row["Counter_Value"])
            cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


if __name__ == "__main__":
    fetch, n1, write = {}, {}, {}
    for a in range(1, len(sys.argv) - 1, 2):
        f_, n_ = collect(sys.argv[a], "FETCH_SIZE")
        w_, _ = collect(sys.argv[a + 1], "WRITE_SIZE")
        for k in f_:
            if k not in fetch:
                fetch[k], n1[k] = f_[k], n_[k]
        for k in w_:
            write.setdefault(k, w_[k])
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if not any(s in k for s in ("pair_gemm", "class_gemm", "class_", "gather_sum", "gather_list", "wgrad_gemm", "wgrad_s", "wgrad_h", "bn_", "conv_nbr", "kmap_",
                                    "devoxelize", "voxelize", "trilinear", "table_", "hash_kernel", "devox_")):
            continue
        f_kib, w_kib = fetch.get(k, 0.0), write.get(k, 0.0)
        out[k] = {"hbm_bytes_per_launch": (2.0 * f_kib + w_kib) * 1024.0, "fetch_kib_raw": f_kib, "write_kib": w_kib,
                  "launches_sampled": int(n1.get(k, 0))}
    json.dump(out, open("profiles/traffic.json", "w"), indent=1)
    for k, v in out.items():
        print(f"{k:42s} {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch  ({v['launches_sampled']} launches)")
