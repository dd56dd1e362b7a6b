"""MFMA utilisation per C-ABI entry point from one rocprofv3 PMC pass over the bench workload:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace ... -- python bench.py ...
    python tools/pmc_mfma.py <dir> <steps_in_run>

Units (MI355X_MICROARCH.md, 'rocprofv3 PMC slots' and the per-instruction table): SQ_VALU_MFMA_BUSY_CYCLES counts
cycles an MFMA pipe is busy, summed over the chip's 1024 SIMDs (= 32 per v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE is
the sum over the 8 XCDs of their active cycles, so kernel cycles = GRBM_GUI_ACTIVE / 8.
    MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)
It is the fraction of the dense MFMA peak the kernel's issue stream could reach at the clock it ran at (short
dispatches read high on GRBM_GUI_ACTIVE, see the guide's DVFS note; launches under ~20 us are listed but not trusted)."""
import collections, csv, glob, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_traffic import entry_of  # noqa: E402  (same kernel -> entry map)


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: collections.Counter())
    for r in csv.DictReader(open(f)):
        e = entry_of(r["Kernel_Name"]) or "other"
        agg[e][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            agg[e]["launches"] += 1
    print(f"{'entry point':24s} {'launches/step':>13s} {'kernel Mcycles/step':>20s} {'MFMA busy':>10s}")
    for e, c in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        if cyc <= 0:
            continue
        util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        print(f"{e:24s} {c['launches'] / steps:13.1f} {cyc / steps / 1e6:20.2f} {100 * util:9.1f} %")


if __name__ == "__main__":
    main()
