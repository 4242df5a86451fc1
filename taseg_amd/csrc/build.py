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
SOURCES = ["hash.hip", "coords.hip", "pointops.hip", "conv.hip", "conv_pairs.hip", "bn.hip", "quantize.hip", "stage.hip", "image.hip", "conv_pairs_h.hip", "conv_pairs_s.hip", "conv_class.hip", "optim.hip", "rccl.hip", "block.hip", "loss.hip", "evaltail.hip", "conv2d_rows.hip", "shuffle_cat.hip"]
# measured-and-shelved kernels (HISTORY.md section 3.1 step 11) live in tools/experiments/ and are NOT part of libtaseg_hip.so:
# `python -m taseg_amd.csrc.build --experiments` builds them into tools/experiments/build/libtaseg_exp.so for the probes there
EXP_DIR = os.path.join(ROOT, "tools", "experiments")
EXPERIMENTS = ["conv_os.hip", "conv_pairs_x.hip"]
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
    return LIB


def build_experiments(verbose=True):
    """tools/experiments/*.hip -> tools/experiments/build/libtaseg_exp.so (links the product library's hash.o for ts_set_error)"""
    build(verbose=verbose)
    out_dir = os.path.join(EXP_DIR, "build")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for name in EXPERIMENTS:
        src, obj = os.path.join(EXP_DIR, name), os.path.join(out_dir, name.replace(".hip", ".o"))
        cmd = [HIPCC, *FLAGS, "-I", HERE, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    out = os.path.join(out_dir, "libtaseg_exp.so")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, os.path.join(OBJ_DIR, "hash.o"),
                    os.path.join(OBJ_DIR, "conv_pairs.o"), "-ldl"], check=True)
    return out


if __name__ == "__main__":
    print(build_experiments() if "--experiments" in sys.argv else build(force="--force" in sys.argv))
