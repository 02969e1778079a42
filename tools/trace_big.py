"""The long kernels of the LAST fused call in a rocprofv3 --kernel-trace CSV (everything after the last but-N covariance build):
  python tools/trace_big.py <kernel_trace.csv> [min_us=300]
start offset, duration, short name, workgroups, queue -- enough to see where a recursive / blocked schedule spends its time."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
def short(n):
    if 'leaf' in n: return 'leaf'
    m = re.search(r'pg_gemm_kernel<\w+, (\w+), (\w+), (\d+), (\d+)', n)
    if m: return 'g%s%s_%sx%s' % ('T' if m.group(1) == 'true' else 'N', 'T' if m.group(2) == 'true' else 'N', m.group(3), m.group(4))
    m = re.search(r'(\w+_kernel|kbuild|copyBuffer|fillBuffer)', n)
    return m.group(1) if m else n[:24]
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']),
             int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r['Queue_Id']) for r in rows)
# calls are separated by host synchronisations: cut where the device idles for > 0.2 ms and take the last segment with >= 50 kernels
segs, cur, hi = [], [], 0
for k in ks:
    if cur and k[0] - hi > 200000:
        segs.append(cur); cur = []
    cur.append(k); hi = max(hi, k[1])
segs.append(cur)
seq = [s for s in segs if len(s) >= 50][-1]
t0 = seq[0][0]
print("last call: %.3f ms, %d kernels" % ((max(k[1] for k in seq) - t0) / 1e6, len(seq)))
for k in seq:
    if (k[1] - k[0]) / 1e3 >= min_us or 'kbuild' in k[2]:
        print("t=%9.3f ms  dur=%9.3f ms  %-22s wgs=%6d q=%s" % ((k[0] - t0) / 1e6, (k[1] - k[0]) / 1e6, k[2], k[3], k[4]))
