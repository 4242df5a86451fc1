mkdir -p gpurun_out
for a in 0 1 2 3 4 8 7 15; do
  TASEG_PG_ABLATE=$a timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/abl_$a.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/abl_$a.json"))
ks={k["kernel"]:k for k in d["kernels"]}
print("ablate=$a", "step %.1f ms"%d["ms_per_step"], " ".join("%s:%.2f"%(n.replace("pair_gemm_kernel",""),ks[n]["ms_per_step"]) for n in ks if "pair_gemm" in n))
PY
done
