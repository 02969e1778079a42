"""Chain-level view of one pg_potrf call from a rocprofv3 --kernel-trace CSV (the LAST call in the trace, i.e. everything
after the last covariance build):  python tools/trace_chain.py <kernel_trace.csv> [leaves_per_panel]
For every outer panel of the panel stream: its span, the time inside leaves / U / T / Sa, and the idle gaps between
kernels of that stream; then the update stream's kernels."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lpp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
def short(n):
    if 'leaf' in n: return 'leaf'
    m = re.search(r'pg_gemm_kernel<\w+, (\w+), (\w+), (\d+), (\d+)', n)
    if m: return 'g%s%s_%sx%s' % ('T' if m.group(1) == 'true' else 'N', 'T' if m.group(2) == 'true' else 'N', m.group(3), m.group(4))
    m = re.search(r'(\w+_kernel|kbuild|copyBuffer|fillBuffer)', n)
    return m.group(1) if m else n[:24]
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']),
             int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r['Queue_Id']) for r in rows)
idx = [i for i, k in enumerate(ks) if 'kbuild' in k[2]]
seq = ks[idx[-1] + 1:]
t0 = seq[0][0]
end = max(k[1] for k in seq)
print("total %.3f ms, kernels per queue:" % ((end - t0) / 1e6), dict(collections.Counter(k[4] for k in seq)))
leafq = collections.Counter(k[4] for k in seq if k[2] == 'leaf').most_common(1)[0][0]
chain = [k for k in seq if k[4] == leafq]
us = lambda ns: ns / 1e3
pan, cur, nl = [], [], 0
for k in chain:
    if k[2] == 'leaf' and nl and nl % lpp == 0 and cur and cur[-1][2] != 'leaf' and not any(c[2] == 'leaf' for c in cur[-1:]):
        pass
    cur.append(k)
    if k[2] == 'leaf': nl += 1
# split into panels: a panel = lpp leaves; boundary = first kernel after the T that follows the lpp-th leaf is Sa (big grid)
pan, cur, nl = [], [], 0
for i, k in enumerate(chain):
    cur.append(k)
    if k[2] == 'leaf':
        nl += 1
    if nl == lpp and (i + 1 == len(chain) or chain[i + 1][2] == 'leaf' or (chain[i + 1][3] > 0 and k[2] != 'leaf' and chain[i + 1][2] != k[2] and cur.count(k) and sum(1 for c in cur if c[2] == 'leaf') == lpp and chain[i + 1][2].endswith(('64x64', '128x128')) and k[2] not in ('leaf',))):
        pass
# simpler: cut after every lpp-th leaf's following non-leaf kernels up to (and including) the first 64x64 / 128x128 kernel
pan, cur, nl, closing = [], [], 0, False
for k in chain:
    if closing and k[2] == 'leaf':
        pan.append(cur); cur = []; nl = 0; closing = False
    if closing and (k[2].endswith('64x64') or k[2].endswith('128x128')) and any(c[2].endswith('64x64') or c[2].endswith('128x128') for c in cur[-1:]):
        pass
    cur.append(k)
    if k[2] == 'leaf':
        nl += 1
        if nl == lpp: closing = True
if cur: pan.append(cur)
print("panel stream (queue %s): per panel  start  span | leaf  U  T  other | idle gaps   (us)" % leafq)
for o, p in enumerate(pan):
    s, e = p[0][0], p[-1][1]
    tot = lambda f: sum(k[1] - k[0] for k in p if f(k))
    gaps = sum(max(0, p[i + 1][0] - p[i][1]) for i in range(len(p) - 1))
    isU = lambda k: k[2] in ('gNT_32x64', 'gNT_32x32') or (k[2] == 'gNT_64x64' and k[3] and False)
    isT = lambda k: k[2] in ('gNT_32x128', 'gNT_64x128')
    print("  %2d: %9.1f %8.1f | %7.1f %7.1f %7.1f %8.1f | %7.1f   n=%d" % (
        o, us(s - t0), us(e - s), us(tot(lambda k: k[2] == 'leaf')), us(tot(isU)), us(tot(isT)),
        us(tot(lambda k: k[2] != 'leaf' and not isU(k) and not isT(k))), us(gaps), len(p)))
for q in set(k[4] for k in seq):
    if q == leafq: continue
    print("queue", q)
    for k in seq:
        if k[4] == q and (k[1] - k[0]) > 20000:
            print("  %9.1f -> %9.1f  %8.1f us  %s wgs=%d" % (us(k[0] - t0), us(k[1] - t0), us(k[1] - k[0]), k[2], k[3]))
if len(sys.argv) > 3:      # dump the panel stream kernel by kernel
    prev = None
    for k in chain:
        print("  %9.1f +%6.1f %-12s wgs=%-5d gap %.1f" % (us(k[0] - t0), us(k[1] - k[0]), k[2], k[3], us(k[0] - prev) if prev else 0))
        prev = k[1]
