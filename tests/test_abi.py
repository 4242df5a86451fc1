"""The C-ABI library loads (no GPU needed) and exports every symbol include/taseg_hip.h declares;
the ctypes table in taseg_amd/_lib.py covers exactly those symbols.  No compute calls here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "taseg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ts_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_reference_boundary():
    syms = _header_symbols()
    # the ten logical entry points of torchsparse.backend (pybind_cuda.cpp:18-39)
    for name in ("ts_hash", "ts_kernel_hash", "ts_hash_query", "ts_count", "ts_voxelize_forward",
                 "ts_voxelize_backward", "ts_devoxelize_forward", "ts_devoxelize_backward",
                 "ts_convolution_forward", "ts_convolution_backward"):
        assert name in syms


def test_library_exports_every_declared_symbol():
    from taseg_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run `python -m taseg_amd.csrc.build` (hipcc cross-compiles without a GPU)"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_symbols():
        assert hasattr(lib, name), f"{name} declared in taseg_hip.h but not exported"


def test_ctypes_table_matches_header():
    from taseg_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_symbols()
    lib = _lib.load()
    assert lib.ts_version().decode().startswith("taseg_hip")


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(ROOT, "include", "taseg_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)          # declarations only, comments stripped
    assert "at::" not in code and "torch" not in code and "Tensor" not in code


def test_product_does_not_import_the_oracle():
    """the oracle is test infrastructure: nothing under taseg_amd/ may import or call it"""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "taseg_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "ts_oracle" in src:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_cpu_tensors_fail_loudly():
    import torch
    from taseg_amd import backend as B
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        B.hash_cuda(torch.zeros((3, 4), dtype=torch.int32))
    with pytest.raises(RuntimeError):
        B.conv_nbr(torch.zeros(3, 4), torch.zeros(27, 4, 8), torch.zeros((27, 3), dtype=torch.int32), 3)


def test_ctypes_structs_have_the_headers_layout(tmp_path):
    """TsConvBlockOpts / TsClassPlan / TsPlaneJob cross the boundary by pointer: the ctypes mirrors in taseg_amd/_lib.py
    (and taseg_amd/planes.py) must have the size and the field offsets a C compiler gives the header's structs."""
    import shutil
    import subprocess
    from taseg_amd import _lib, planes
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    structs = {"TsConvBlockOpts": _lib.TsConvBlockOpts, "TsClassPlan": _lib.TsClassPlan, "TsPlaneJob": planes.TsPlaneJob}
    lines = []
    for name, st in structs.items():
        offs = ", ".join(f"offsetof({name}, {f[0]})" for f in st._fields_)
        fmt = " ".join(["%zu"] * (1 + len(st._fields_)))
        lines.append(f'  printf("{name} {fmt}\\n", sizeof({name}), {offs});')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "taseg_hip.h"\nint main(void) {\n' + "\n".join(lines) +
                   "\n  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()
    seen = {ln.split()[0]: [int(v) for v in ln.split()[1:]] for ln in out}
    for name, st in structs.items():
        mine = [ctypes.sizeof(st)] + [getattr(st, f[0]).offset for f in st._fields_]
        assert seen[name] == mine, (name, seen[name], mine)
