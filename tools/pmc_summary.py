"""Sum FETCH_SIZE / WRITE_SIZE (KiB per dispatch) over the kernels of the LAST evaluation in two rocprofv3 --pmc
output directories (see probe_eval_once.py).  gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section):
FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads -> doubled."""
import csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd import _lib
def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    last = max(i for i, r in enumerate(rows) if "kbuild" in r["Kernel_Name"] and "grad" not in r["Kernel_Name"])
    return rows[last:]
def group(name):
    if "pg_gemm_kernel" in name or "pg_gemm_mixed_kernel" in name: return "gemm_core"
    m = re.search(r"(pg_\w+|\w+_kernel)", name)
    return m.group(1) if m else name[:30]
out = {}
for d, c in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    for r in load(d, c):
        g = out.setdefault(group(r["Kernel_Name"]), {"launches": 0, "FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0})
        g[c + "_KiB"] += float(r["Counter_Value"])
        if c == "FETCH_SIZE": g["launches"] += 1
for g in out.values():
    g["hbm_bytes"] = (2.0 * g["FETCH_SIZE_KiB"] + g["WRITE_SIZE_KiB"]) * 1024.0
    g["hbm_bytes_per_launch"] = g["hbm_bytes"] / max(1, g["launches"])
print(json.dumps({"build": _lib.build_id(), "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/probe_eval_once.py; "
                            "one NLML+grad evaluation at N=16384; FETCH_SIZE doubled (gfx950 correction)", "kernels": out}, indent=1))
