"""Build libtaseg_hip.so (gfx950) in-tree with hipcc.  No torch / cmake involved.

    python -m taseg_amd.csrc.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the resulting shared library sits
next to the package (taseg_amd/libtaseg_hip.so) so it travels with the tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
SOURCES = ["hash.hip", "coords.hip", "pointops.hip", "conv.hip", "conv_pairs.hip", "bn.hip", "quantize.hip", "image.hip", "conv_pairs_h.hip", "conv_pairs_s.hip", "conv_os.hip", "conv_class.hip", "optim.hip", "rccl.hip", "block.hip", "loss.hip"]
EXPERIMENTS = {"conv_pairs_x.hip": "libtaseg_x.so"}      # three-product IEEE-half split pair GEMM (tools/x_probe.py)
LIB = os.path.join(PKG, "libtaseg_hip.so")
OBJ_DIR = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# files whose float results must equal numpy's bit for bit (pose fuse, voxel rounding): hipcc's default
# -ffp-contract=fast would fuse a*b+c into one rounding
EXTRA = {"pointops.hip": ["-ffp-contract=off"], "quantize.hip": ["-ffp-contract=off"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(ROOT, "include")]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def _deps():
    return [os.path.join(HERE, "common.h"), os.path.join(ROOT, "include", "taseg_hip.h"), os.path.abspath(__file__)]


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(HERE, s)
        obj = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(d, obj) for d in _deps()):
            jobs.append([HIPCC, *FLAGS, *EXTRA.get(s, []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, flush=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"])
    # measured-and-shelved kernels: their own small library next to the objects, loaded only by tools/
    for src_name, lib_name in EXPERIMENTS.items():
        src, obj = os.path.join(HERE, src_name), os.path.join(OBJ_DIR, src_name.replace(".hip", ".o"))
        out = os.path.join(OBJ_DIR, lib_name)
        if os.path.exists(src) and (force or _newer(src, out) or any(_newer(d, out) for d in _deps())):
            run([HIPCC, *FLAGS, "-c", src, "-o", obj])
            run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj, os.path.join(OBJ_DIR, "hash.o"), "-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
