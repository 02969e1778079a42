"""FETCH_SIZE / WRITE_SIZE (KiB) of the LAST GEMM-core dispatch in a rocprofv3 --pmc output directory (FETCH doubled per the gfx950 note)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pg_gemm" in r["Kernel_Name"]]
last = max(int(r["Dispatch_Id"]) for r in rows)
for r in rows:
    if int(r["Dispatch_Id"]) == last:
        v = float(r["Counter_Value"]) * 1024.0 * (2.0 if r["Counter_Name"] == "FETCH_SIZE" else 1.0)
        print("%s of the last GEMM-core launch: %.2f GB%s" % (r["Counter_Name"], v / 1e9, " (doubled)" if r["Counter_Name"] == "FETCH_SIZE" else ""))
