"""Per-family kernel time of a TIAF step from a rocprofv3 --kernel-trace --stats run (the *_kernel_stats.csv of
`rocprofv3 ... -- python3 bench.py --workload tiaf --amp`): kernels grouped by the branch of the step they belong to, by name.

    python tools/tiaf_families.py <kernel_stats.csv> <steps in the trace>
"""
import csv
import re
import sys

FAMILIES = [
    ("UNet2D convolutions (ours: csrc/conv2d_rows.hip - 3x3 up to 96 input channels, 1x1 + LeakyReLU, weight + bias gradients)", r"conv3x3|conv1x1"),
    ("UNet2D LeakyReLU + BatchNorm2d as one node (ours: csrc/bn.hip lbn_*)", r"lbn_"),
    ("UNet2D PixelShuffle + dropout + concat (ours: csrc/shuffle_cat.hip)", r"shuffle_cat"),
    ("vendor libraries (MIOpen / hipBLASLt / CK): UNet2D layers at 1/4 scale and below, input layer, classifier; the fusion head's GEMMs", r"igemm_|^Cijk_|Custom_Cijk|miopenSp3|grouped_conv|naive_conv|batched_transpose|SubTensorOp"),
    ("UNet2D BatchNorm2d (MIOpen / ATen)", r"MIOpenBatchNorm|batch_norm_"),
    ("ATen kernels of the whole step (residual adds, gradient accumulation, LeakyReLU of the wide layers, casts, concatenations, index glue)", r"at::native::|at_cuda_detail"),
    ("UNet2D average pooling (ours, channels-last)", r"avgpool3s2"),
    ("image gather + adjoint + plan (ours)", r"image_"),
    ("losses: CE + Lovasz (ours) and their sorts", r"lovasz|softmax_ce|ce_lovasz"),
    ("sparse convolution: class / pair GEMM, pass 2, weight gradient (ours)", r"class_gemm|pair_gemm|gather_list|gather_sum|wgrad|conv_nbr|split_planes|cast_weights|cat_cols|copy_cols"),
    ("sparse BatchNorm + activation (ours)", r"bn_"),
    ("voxel <-> point transfer (ours)", r"devox|voxelize|trilinear"),
    ("index plans: hash tables, kernel maps, class plans (ours)", r"kmap_|table_|class_keys|class_fill|class_tiles|ds_pack|ds_unpack|uq_|fill_segments|nbr_"),
    ("data stage: fuse, project, voxelise, quantise (ours)", r"fuse_|project_|vc_|sq_|seg_min|stage_|quant"),
    ("rocPRIM sorts / scans (index plans, loss, data stage)", r"rocprim"),
    ("optimizer (ours)", r"sgd_"),
    ("runtime fills / copies", r"__amd_rocclr"),
]


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    rows = list(csv.DictReader(open(path)))
    acc = {name: [0.0, 0] for name, _ in FAMILIES}
    acc["other"] = [0.0, 0]
    other = []
    for r in rows:
        ms, calls = float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps
        for name, pat in FAMILIES:
            if re.search(pat, r["Name"]):
                acc[name][0] += ms
                acc[name][1] += calls
                break
        else:
            acc["other"][0] += ms
            acc["other"][1] += calls
            other.append((ms, r["Name"][:100]))
    total = sum(v[0] for v in acc.values())
    print(f"kernel time per step {total:.2f} ms in {sum(v[1] for v in acc.values()):.0f} launches (all streams; {steps:.0f} steps in the trace)")
    for name, (ms, calls) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print(f"{ms:8.2f} ms  {100 * ms / total:5.1f} %  {calls:7.1f} launches  {name}")
    for ms, n in sorted(other, reverse=True)[:8]:
        print(f"   other: {ms:.3f} ms  {n}")


if __name__ == "__main__":
    main()
