"""Build oracle/_ref/ts_ref_backend.so from the REFERENCE's own C++ sources (build container only).

Recipe (ours, not the reference's setup.py): the CPU kernel files of torchsparse v1.4.0 are
extracted from /root/reference/package/torchsparse.zip into a temporary directory, compiled
in place with g++ against the installed libtorch headers (flags of TS/setup.py:25-28: -O3 -fopenmp),
and linked with oracle/ref_bind.cpp.  Only the resulting .so lands in oracle/_ref/ (git-ignored,
travels to the GPU box as a built artefact).  No reference source is copied into the repo.

Not built: others/query_cpu.cpp (needs sparsehash's configure-generated header; treated as
unbuildable).  The hash-query semantics are restated in oracle/ts_oracle.py::sphashquery.
"""
import os
import shutil
import subprocess
import sys
import tempfile
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
ZIP = "/root/reference/package/torchsparse.zip"
FILES = ["hash/hash_cpu.cpp", "others/count_cpu.cpp", "voxelize/voxelize_cpu.cpp",
         "devoxelize/devoxelize_cpu.cpp", "convolution/convolution_cpu.cpp"]
NAME = "ts_ref_backend"


def build(force=False, verbose=True):
    out = os.path.join(OUT_DIR, NAME + ".so")
    if os.path.exists(out) and not force:
        return out
    if not os.path.exists(ZIP):
        if verbose:
            print("build_ref: /root/reference not present - keeping prebuilt oracle/_ref (if any)")
        return out if os.path.exists(out) else None
    import sysconfig
    import torch
    from torch.utils.cpp_extension import include_paths
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="ts_ref_build_")
    try:
        with zipfile.ZipFile(ZIP) as z:
            z.extractall(tmp)
        src = os.path.join(tmp, "torchsparse", "torchsparse", "backend")
        incs = []
        for p in include_paths() + [sysconfig.get_paths()["include"], src]:
            incs += ["-I", p]
        libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
        cmd = ["g++", "-O3", "-fopenmp", "-fPIC", "-shared", "-std=c++17", "-w",
               f"-DTORCH_EXTENSION_NAME={NAME}", "-DTORCH_API_INCLUDE_EXTENSION_H",
               f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", *incs,
               os.path.join(HERE, "ref_bind.cpp"), *[os.path.join(src, f) for f in FILES],
               "-L", libdir, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", f"-Wl,-rpath,{libdir}",
               "-o", out]
        if verbose:
            print("build_ref:", " ".join(cmd[:12]), "...", flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle/_ref build failed:\n" + r.stdout + r.stderr)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
