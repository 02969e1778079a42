"""Kernel time by kind for the LAST call in a rocprofv3 --kernel-trace CSV (calls are separated by device-idle gaps > 0.2 ms):
per short kernel name: launches, total busy ms, share; plus the call's span and the per-queue busy time."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    if 'leaf' in n: return 'leaf'
    m = re.search(r'pg_gemm_(mixed_)?kernel<(\w+)(?:, (\w+), (\w+), (\d+), (\d+))?', n)
    if m and m.group(1): return 'gMIXED'
    if m: return 'g%s%s_%sx%s' % ('T' if m.group(3) == 'true' else 'N', 'T' if m.group(4) == 'true' else 'N', m.group(5), m.group(6))
    m = re.search(r'(\w+_kernel|kbuild|copyBuffer|fillBuffer)', n)
    return m.group(1) if m else n[:24]
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r['Queue_Id']) for r in rows)
segs, cur, hi = [], [], 0
for k in ks:
    if cur and k[0] - hi > 200000:
        segs.append(cur); cur = []
    cur.append(k); hi = max(hi, k[1])
segs.append(cur)
minn = int(sys.argv[2]) if len(sys.argv) > 2 else 30
seq = [s for s in segs if len(s) >= minn][-1]
t0, t1 = seq[0][0], max(k[1] for k in seq)
print("call: %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(seq)))
agg = collections.OrderedDict()
for k in seq:
    a = agg.setdefault((k[2], k[4]), [0, 0])
    a[0] += 1; a[1] += k[1] - k[0]
for (name, q), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-26s q=%s launches=%4d busy=%8.3f ms (%.0f %% of the span)" % (name, q, c, t / 1e6, 100.0 * t / (t1 - t0)))
