"""profiles/*_counters.csv (tools/profile_bench.sh) -> a table of matrix-pipe utilisation and fabric traffic per kernel."""
import csv, sys

out = ["# Matrix-pipe utilisation and fabric traffic from rocprofv3 counters", "",
       "One `--pmc` pass each for FETCH_SIZE, WRITE_SIZE and the SQ / GRBM counters (`tools/profile_bench.sh`), values per launch.",
       "busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): share of SIMD-cycles, at the clock actually held, with the",
       "matrix pipe busy.  read = FETCH_SIZE x 2 (the gfx950 correction of the microarchitecture guide), write = WRITE_SIZE, in MB.", "",
       "| workload | kernel | launches | cycles per XCD | matrix pipe busy | waves waiting | read MB | write MB |", "|---|---|---|---|---|---|---|---|"]
for path, what in zip(sys.argv[1::2], sys.argv[2::2]):
    for r in csv.DictReader(open(path)):
        mf, gui = float(r["SQ_VALU_MFMA_BUSY_CYCLES"]), float(r["GRBM_GUI_ACTIVE"])
        if gui <= 0 or (mf == 0 and float(r["read_MB(FETCH_SIZE x2)"]) < 100):
            continue
        wait = float(r["SQ_WAIT_INST_ANY"]) / max(float(r["SQ_WAVE_CYCLES"]), 1.0)
        out.append(f"| {what} | `{r['kernel']}` | {r['launches']} | {gui / 8:,.0f} | {100 * mf / (1024 * gui / 8):.0f} % | {100 * wait:.0f} % | "
                   f"{float(r['read_MB(FETCH_SIZE x2)']):.0f} | {float(r['write_MB']):.0f} |")
print("\n".join(out))
