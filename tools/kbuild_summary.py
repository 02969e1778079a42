"""rocprofv3 evidence for the covariance build (tools/probe_kbuild_once.py): duration from the kernel trace, FETCH_SIZE / WRITE_SIZE from the
two --pmc passes (FETCH_SIZE doubled: gfx950 correction, MI355X_MICROARCH.md), against the algorithmic bytes 8 n^2 (4 n (n + 64) lower-only)
+ 8 n d.  Dispatches 1-3 are lower-only builds, 4-6 mirrored ones (pg_kbuild_kernel<double, false|true, 2>)."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd import _lib
n, d = 16384, 8
def rows(dirn, pat):
    f = glob.glob(dirn + "/**/*" + pat, recursive=True)[0]
    return [r for r in csv.DictReader(open(f)) if "pg_kbuild_kernel" in r.get("Kernel_Name", "")]
tr = rows(sys.argv[1], "kernel_trace.csv")
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr]
def counter(dirn, name):
    rs = [r for r in rows(dirn, "counter_collection.csv") if r["Counter_Name"] == name]
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) * 1024.0 for r in rs]
fe, wr = counter(sys.argv[2], "FETCH_SIZE"), counter(sys.argv[3], "WRITE_SIZE")
out = {"build": _lib.build_id(), "peak_GBs": 8000.0, "n": n, "d": d, "source": "rocprofv3 --kernel-trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE (three passes) -- python3 tools/probe_kbuild_once.py"}
for name, sl, alg in (("lower_only", slice(0, 3), 4.0 * n * (n + 64) + 8.0 * n * d), ("mirrored", slice(3, 6), 8.0 * n * n + 8.0 * n * d)):
    ms = min(dur[sl])
    out[name] = {"ms_min_of_3": ms, "ms_all": dur[sl], "algorithmic_bytes": alg, "achieved_GBs": alg / ms / 1e6, "frac_of_peak": alg / ms / 1e6 / 8000.0,
                 "hbm_bytes_pmc": 2.0 * fe[sl][-1] + wr[sl][-1], "fetch_bytes_x2": 2.0 * fe[sl][-1], "write_bytes": wr[sl][-1],
                 "traffic_over_algorithmic": (2.0 * fe[sl][-1] + wr[sl][-1]) / alg}
print(json.dumps(out, indent=1))
