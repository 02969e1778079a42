"""Where an NLML+gradient evaluation's wall time goes beyond its big kernels: from a rocprofv3 --kernel-trace CSV of a loop of
evaluations (tools/probe_eval_loop.py), per evaluation: span first kernel -> last kernel, idle gap to the next evaluation's first
kernel, and the time of the named phases on the caller's queue."""
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
# an evaluation ends with pg_grad_reduce_kernel
ends = [i for i, k in enumerate(ks) if 'grad_reduce' in k[2]]
prev_end = None
for e in range(1, len(ends)):
    seg = ks[ends[e - 1] + 1: ends[e] + 1]
    t0, t1 = seg[0][0], max(k[1] for k in seg)
    big = collections.OrderedDict()
    for k in seg:
        if (k[1] - k[0]) > 2e6: big.setdefault(k[2] + "/%d" % k[3], []).append((k[0] - t0, k[1] - k[0]))
    gap = (t0 - prev_end) / 1e3 if prev_end else 0.0
    print("evaluation %d: span %.3f ms, idle before it %.1f us, kernels %d" % (e, (t1 - t0) / 1e6, gap, len(seg)))
    for name, v in big.items():
        print("     %-22s %s" % (name, "  ".join("@%.2f +%.3f" % (a / 1e6, b / 1e6) for a, b in v)))
    prev_end = t1
