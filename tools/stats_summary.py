"""Readable summary of a rocprofv3 *_kernel_stats.csv, plus the GEMM core aggregated over its instantiations."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
gc = gt = 0
for r in rows:
    name = r["Name"].replace("void ", "")
    if "pg_gemm_kernel" in name or "pg_gemm_mixed_kernel" in name:
        gc += int(r["Calls"]); gt += int(r["TotalDurationNs"])
    print("%-78s calls=%5d total_ms=%9.2f avg_us=%9.1f pct=%6.2f" % (name[:78], int(r["Calls"]), int(r["TotalDurationNs"]) / 1e6,
                                                                  float(r["AverageNs"]) / 1e3, 100.0 * int(r["TotalDurationNs"]) / tot))
print("GEMM core, all instantiations: calls=%d total_ms=%.2f avg_ms=%.4f (%.1f %% of kernel time)" % (gc, gt / 1e6, gt / 1e6 / max(gc, 1), 100.0 * gt / tot))
