"""Build the optional native autograd binding (taseg_amd/_fast_block.so): plain C++ against the installed PyTorch,
no device code (g++ only; the kernels stay in libtaseg_hip.so, bound with dlopen at import time).

    python -m taseg_amd.csrc.fastpath.build [--force]
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(os.path.dirname(HERE))
ROOT = os.path.dirname(PKG)
NAME = "_fast_block"
OUT = os.path.join(PKG, NAME + ".so")
SRC = os.path.join(HERE, "fast_block.cpp")


def build(force=False, verbose=False):
    deps = [SRC, os.path.join(HERE, "stage_program.h"), os.path.join(ROOT, "include", "taseg_hip.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    from torch.utils import cpp_extension
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    mod = cpp_extension.load(name=NAME, sources=[SRC], extra_include_paths=[os.path.join(ROOT, "include")],
                             extra_cflags=["-O2", "-std=c++17"], extra_ldflags=["-ldl"], build_directory=bdir,
                             with_cuda=False, verbose=verbose)
    shutil.copyfile(mod.__file__, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
