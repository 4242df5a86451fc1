"""SQ counters of the 2-D convolution kernels (csrc/conv2d_rows.hip) from a rocprofv3 --pmc run of tools/conv2d_probe.py:

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY \\
        SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d <dir> -o p -- python3 tools/conv2d_probe.py general
    python tools/pmc_conv2d.py <dir>/.../p_counter_collection.csv

Per kernel: mean counter values per launch and the shares of a wave's cycles (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count
quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles: MI355X_MICROARCH.md)."""
import collections
import csv
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        n = r["Kernel_Name"]
        if "conv3x3" in n or "conv1x1" in n:
            acc[n.split("(")[0][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, d in sorted(acc.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        print(n)
        for c, v in sorted(m.items()):
            print("   %-28s mean per launch %.4g  (%d launches)" % (c, v, len(d[c])))
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        if wc > 0:
            print("   of the waves' cycles: parked (s_waitcnt / barrier) %.0f %%, issue stalls %.0f %% (LDS issue %.0f %%), issuing %.0f %%; "
                  "MFMA pipe busy %.0f %% of wave cycles, LDS bank-conflict cycles %.1f %% of LDS-active"
                  % (100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_LDS", 0) / wc,
                     100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * wc),
                     100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 0), 1.0)))


if __name__ == "__main__":
    main()
