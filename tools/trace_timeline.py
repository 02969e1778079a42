"""Per-panel timeline of one pg_potrf / pg_potrf_trtri call from a rocprofv3 --kernel-trace CSV
(python tools/trace_timeline.py <kernel_trace.csv>): chain spans, update kernels, background kernels."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    if 'leaf' in n: return 'leaf'
    m = re.search(r'pg_gemm_kernel<double, (\w+), (\w+), (\d+), (\d+)', n)
    if m: return 'g%s%s_%sx%s' % ('T' if m.group(1) == 'true' else 'N', 'T' if m.group(2) == 'true' else 'N', m.group(3), m.group(4))
    m = re.search(r'(\w+_kernel|kbuild|copyBuffer|fillBuffer)', n)
    return m.group(1) if m else n[:24]
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']),
             int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r['Queue_Id']) for r in rows)
idx = [i for i, k in enumerate(ks) if 'kbuild' in k[2]]
seq = ks[idx[-1] + 1:]
t0 = seq[0][0]
qs = collections.Counter(k[4] for k in seq)
print("queues:", dict(qs), " total %.3f ms" % ((max(k[1] for k in seq) - t0) / 1e6))
leafq = collections.Counter(k[4] for k in seq if k[2] == 'leaf').most_common(1)[0][0]
chain = [k for k in seq if k[4] == leafq]
panels, cur, nl = [], [], 0
for k in chain:
    cur.append(k)
    if k[2] == 'leaf': nl += 1
    if k[2].endswith('64x128') and nl % 8 == 0 and len(panels) < nl // 8:
        panels.append(cur); cur = []
if cur: panels.append(cur)
for o, p in enumerate(panels):
    s, e = p[0][0], p[-1][1]
    tot = lambda f: sum(k[1] - k[0] for k in p if f(k)) / 1e6
    print("chain %2d: %8.3f -> %8.3f span %6.3f  leaf %.3f U %.3f T %.3f other %.3f" % (
        o, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, tot(lambda k: k[2] == 'leaf'), tot(lambda k: k[2].endswith('64x64')),
        tot(lambda k: k[2].endswith('64x128')), tot(lambda k: k[2] != 'leaf' and not k[2].endswith('64x64') and not k[2].endswith('64x128'))))
for q in qs:
    if q == leafq: continue
    print("queue", q)
    for k in seq:
        if k[4] == q and (k[1] - k[0]) > 30000:
            print("  %8.3f -> %8.3f  %6.3f ms  %s wgs=%d" % ((k[0] - t0) / 1e6, (k[1] - t0) / 1e6, (k[1] - k[0]) / 1e6, k[2], k[3]))
