"""Chip-utilisation timeline of the LAST factorisation in a rocprofv3 --kernel-trace CSV, joined with the launch log the library
writes under PG_GEMM_LOG (flops of every GEMM launch, in launch order):
    python tools/trace_util.py <kernel_trace.csv> <gemm_log> [bucket_ms]
Per time bucket: TFLOP/s delivered by each queue's GEMMs (a launch's flops spread evenly over its duration), the leaf time of
the panel queue, and the number of GEMM launches in flight."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
log = [l.split() for l in open(sys.argv[2]) if l.strip()]
bucket = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
VAR = {0: ("false", "true", 128, 128), 1: ("false", "true", 64, 256), 2: ("false", "false", 128, 128), 3: ("true", "false", 128, 128),
       4: ("false", "false", 128, 128), 5: ("true", "true", 128, 128), 6: ("false", "true", 64, 64), 7: ("false", "true", 64, 128),
       8: ("false", "true", 32, 64), 9: ("false", "true", 32, 128), 10: ("true", "true", 64, 64), 11: ("false", "true", 32, 32)}
def key_of_log(l):
    v, sz, w = int(l[0]), int(l[1]), int(l[2])
    ta, tb, bm, bn = VAR[v]
    nw = w if (bm, bn) == (128, 128) else 4
    return ("double" if sz == 8 else "float", ta, tb, bm, bn, 1 if v == 4 else 0, nw)
def key_of_name(n):
    m = re.search(r'pg_gemm_kernel<(\w+), (\w+), (\w+), (\d+), (\d+), (\d+), (\d+), (\d+)>', n)
    return (m.group(1), m.group(2), m.group(3), int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(8))) if m else None
# per kernel-name key: flops in launch order; trace rows of a key in dispatch order
byk = collections.defaultdict(list)
for l in log: byk[key_of_log(l)].append(float(l[7]))
ks = sorted(rows, key=lambda r: int(r['Dispatch_Id']))
cnt = collections.Counter()
ev = []
for r in ks:
    k = key_of_name(r['Kernel_Name'])
    s, e, q = int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id']
    if k is None:
        ev.append((s, e, q, 0.0, 'leaf' if 'leaf' in r['Kernel_Name'] else ('kbuild' if 'kbuild' in r['Kernel_Name'] else 'other')))
        continue
    i = cnt[k]; cnt[k] += 1
    fl = byk[k][i] if i < len(byk[k]) else 0.0
    ev.append((s, e, q, fl, 'gemm'))
for k in byk:
    if cnt[k] != len(byk[k]): print("warning: %s: %d trace rows vs %d log lines" % (k, cnt[k], len(byk[k])))
builds = [i for i, x in enumerate(ev) if x[4] == 'kbuild']
ev = sorted(ev)
last_build = max(x[0] for x in ev if x[4] == 'kbuild')
seq = [x for x in ev if x[0] > last_build]
t0 = min(x[0] for x in seq); t1 = max(x[1] for x in seq)
qs = sorted(set(x[2] for x in seq))
leafq = collections.Counter(x[2] for x in seq if x[4] == 'leaf').most_common(1)[0][0]
print("total %.3f ms; queues %s; panel queue %s; total GEMM flops %.3e" % ((t1 - t0) / 1e6, qs, leafq, sum(x[3] for x in seq)))
nb = int((t1 - t0) / 1e6 / bucket) + 1
tf = {q: [0.0] * nb for q in qs}; leaf = [0.0] * nb
for s, e, q, fl, kind in seq:
    d = max(1, e - s)
    b0, b1 = int((s - t0) / 1e6 / bucket), int((e - t0) / 1e6 / bucket)
    for b in range(b0, b1 + 1):
        lo, hi = max(s, t0 + b * bucket * 1e6), min(e, t0 + (b + 1) * bucket * 1e6)
        if hi <= lo: continue
        if kind == 'gemm': tf[q][b] += fl * (hi - lo) / d
        if kind == 'leaf': leaf[b] += (hi - lo)
print("  t(ms)  " + "  ".join("q%-5s" % q for q in qs) + "   total TF/s   leaf-busy")
for b in range(nb):
    per = [tf[q][b] / (bucket * 1e-3) / 1e12 for q in qs]
    print("%7.1f  " % (b * bucket) + "  ".join("%6.1f" % v for v in per) + "   %6.1f       %4.0f %%" % (sum(per), 100 * leaf[b] / (bucket * 1e6)))
