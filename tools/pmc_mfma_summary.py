"""MFMA counters of the GEMM-core launches of the LAST call in a rocprofv3 --pmc output directory (tools/probe_mfma_pass.py):
  python tools/pmc_mfma_summary.py <dir> potrf|lauum|fused [n]
SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 = flop the matrix pipe executed (cross-check of the tile-granular count, pg_gemm_flops);
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCC x 1024 SIMDs) = the fraction of SIMD-cycles the matrix pipe was busy
(rocprofv3's MfmaUtil expression; GRBM_GUI_ACTIVE is reported summed over the 8 XCCs)."""
import csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd import _lib
d, what = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
disp = {}
for r in rows:
    e = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(disp)
builds = [i for i in ids if "kbuild" in disp[i]["name"]]
seq = [i for i in ids if i > builds[-1]]
if what == "lauum":      # only the L^-T L^-1 launch: the last GEMM-core dispatch
    seq = [i for i in seq if "pg_gemm" in disp[i]["name"]][-1:]
def grp(name):
    if "pg_gemm" in name: return "gemm_core"
    m = re.search(r"(pg_\w+|\w+_kernel)", name)
    return m.group(1) if m else name[:30]
out = {}
for i in seq:
    e = disp[i]
    g = out.setdefault(grp(e["name"]), {"launches": 0, "SQ_VALU_MFMA_BUSY_CYCLES": 0.0, "SQ_INSTS_VALU_MFMA_MOPS_F64": 0.0, "GRBM_GUI_ACTIVE": 0.0})
    g["launches"] += 1
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "GRBM_GUI_ACTIVE"):
        g[c] += e.get(c, 0.0)
for g in out.values():
    g["mfma_flop"] = 512.0 * g["SQ_INSTS_VALU_MFMA_MOPS_F64"]
    cyc = g["GRBM_GUI_ACTIVE"] / 8.0
    g["gpu_active_cycles_per_xcc"] = cyc
    g["mfma_busy_frac_of_simd_cycles"] = g["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0) if cyc > 0 else None
algo = {"potrf": n ** 3 / 3.0, "lauum": n ** 3 / 3.0, "fused": 2.0 * n ** 3 / 3.0}[what]
gc = out.get("gemm_core", {})
print(json.dumps({"build": _lib.build_id(), "what": what, "n": n,
                  "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE -- python3 tools/probe_mfma_pass.py %s (one kernel at a time: classic chain, single queue)" % what,
                  "algorithmic_flop": algo, "mfma_flop_over_algorithmic": gc.get("mfma_flop", 0.0) / algo if gc else None,
                  "kernels": out}, indent=1))
