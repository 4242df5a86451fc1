"""Diagnostic: registers / spills / occupancy of the kernels of one csrc file whose mangled name contains a pattern.
    python tools/kres.py conv_class.hip class_gemm_h2"""
import os, re, subprocess, sys
src, pat = sys.argv[1], sys.argv[2]
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "taseg_amd", "csrc")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/tmp/kres.o",
                      "-Rpass-analysis=kernel-resource-usage"], cwd=csrc, capture_output=True, text=True).stderr
cur, seen = None, set()
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        continue
    if cur is None:
        continue
    for key, tag in (("VGPRs", "v"), ("AGPRs", "a"), ("VGPRs Spill", "spill"), ("Occupancy [waves/SIMD]", "occ"),
                     ("ScratchSize [bytes/lane]", "scratch")):
        m = re.search(r"\s" + re.escape(key) + r": (\d+)", line)
        if m:
            cur[tag] = int(m.group(1))
    if "LDS Size" in line:
        if pat in cur["name"] and cur["name"] not in seen:
            seen.add(cur["name"])
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            print(f"{name.split('(')[0][:70]:70s} VGPR {cur.get('v'):4d} AGPR {cur.get('a'):4d} spill {cur.get('spill'):4d} "
                  f"scratch {cur.get('scratch'):4d} occ {cur.get('occ')}")
        cur = None
