"""One window of an evaluation, kernel by kernel: from a rocprofv3 --kernel-trace CSV of tools/probe_eval_loop.py, the kernels of
evaluation E that start between the end of its mixed-tile update (the split's A22 -= L21 L21^T) and the start of the next product of
4096 tiles -- i.e. the second half's fused factor-and-invert sub-call -- as (start offset, duration, queue, name/workgroups), with the
busy time per queue, the union of busy intervals and the gaps above 20 us.  Usage: trace_window.py <kernel_trace.csv> [E]"""
import csv, sys, collections
sys.path.insert(0, __file__.rsplit('/', 1)[0])
rows = list(csv.DictReader(open(sys.argv[1])))
E = int(sys.argv[2]) if len(sys.argv) > 2 else 2
import re
def short(n):
    if 'leaf' in n: return 'leaf'
    if 'rowstep' in n: return 'rows'
    m = re.search(r'pg_gemm_(mixed_)?kernel<(\w+)(?:, (\w+), (\w+), (\d+), (\d+))?', n)
    if m and m.group(1): return 'gMIXED'
    if m: return 'g%s%s_%sx%s' % ('T' if m.group(3) == 'true' else 'N', 'T' if m.group(4) == 'true' else 'N', m.group(5), m.group(6))
    m = re.search(r'(\w+_kernel|kbuild|copyBuffer|fillBuffer)', n)
    return m.group(1) if m else n[:24]
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r['Queue_Id']) for r in rows)
ends = [i for i, k in enumerate(ks) if 'grad_reduce' in k[2]]
seg = ks[ends[E - 1] + 1: ends[E] + 1]
t0 = seg[0][0]
mixed = [k for k in seg if k[2] == 'gMIXED'][0]
nxt = [k for k in seg if k[0] > mixed[1] and k[3] == 4096 and k[1] - k[0] > 2e6][0]
win = [k for k in seg if k[0] >= mixed[1] - 1000 and k[0] < nxt[0]]
print("window %.3f .. %.3f ms of evaluation %d (%.3f ms), %d kernels" % ((mixed[1] - t0) / 1e6, (nxt[0] - t0) / 1e6, E, (nxt[0] - mixed[1]) / 1e6, len(win)))
perq = collections.defaultdict(float)
for k in win: perq[k[4]] += (k[1] - k[0]) / 1e6
print("busy per queue (ms):", {q: round(v, 3) for q, v in perq.items()})
iv = sorted((k[0], k[1]) for k in win)
busy, cur0, cur1, gaps = 0.0, iv[0][0], iv[0][1], []
for a, b in iv[1:]:
    if a > cur1:
        busy += cur1 - cur0
        if a - cur1 > 20000: gaps.append(((cur1 - mixed[1]) / 1e6, (a - cur1) / 1e3))
        cur0, cur1 = a, b
    else: cur1 = max(cur1, b)
busy += cur1 - cur0
print("union of busy intervals %.3f ms; gaps > 20 us: %s" % (busy / 1e6, ["@%.3f: %.0f us" % g for g in gaps]))
# runs of the same kernel kind on the same queue
out, last = [], None
for k in win:
    key = (k[2], k[4])
    if last and last[0] == key: last[2] += 1; last[3] += (k[1] - k[0]); last[4] = k[1]
    else:
        last = [key, k[0], 1, k[1] - k[0], k[1], k[3]]
        out.append(last)
for o in out:
    if o[3] > 30000 or o[2] == 1 and o[3] > 15000:
        print("  @%.3f  %-18s q%-3s x%-3d busy %.3f ms  until %.3f  (first launch %d workgroups)" % ((o[1] - mixed[1]) / 1e6, o[0][0], o[0][1], o[2], o[3] / 1e6, (o[4] - mixed[1]) / 1e6, o[5]))
# what else ran while each product of 4096 tiles (and the mixed update) ran: other kernels overlapping its interval
print("kernels overlapping the large products of evaluation %d:" % E)
for big in [k for k in seg if (k[3] == 4096 or k[2] == 'gMIXED') and k[1] - k[0] > 2e6]:
    others = [k for k in seg if k is not big and k[0] < big[1] and k[1] > big[0]]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for k in others:
        a = agg[(k[2], k[4])]; a[0] += 1; a[1] += (min(k[1], big[1]) - max(k[0], big[0])) / 1e6
    print("  %-14s @%.2f +%.3f ms (queue %s): %s" % (big[2], (big[0] - t0) / 1e6, (big[1] - big[0]) / 1e6, big[4],
          ", ".join("%s q%s x%d %.3f ms" % (n, q, c, t) for (n, q), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])) or "nothing"))
