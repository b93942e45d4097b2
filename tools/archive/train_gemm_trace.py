"""Per-shape GEMM time inside a real training forward + backward: joins the library's product log (JODO_TRAIN_GEMM_LOG, call order)
with a rocprofv3 kernel trace of the same process (k_gemm dispatches in start order).
    python tools/train_gemm_trace.py <gemm.log> <kernel_trace.csv> [first N products: those of the step itself, before the isolation loops]"""
import collections, csv, sys
shapes = [tuple(int(v) for v in l.split()[:5]) for l in open(sys.argv[1])]
rows = [r for r in csv.DictReader(open(sys.argv[2])) if 'k_gemm' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = min(len(shapes), len(rows))
if len(sys.argv) > 3:
    n = min(n, int(sys.argv[3]))
agg = collections.defaultdict(lambda: [0, 0.0])
for sh, r in zip(shapes[:n], rows[:n]):
    a = agg[sh]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
print('%d products joined, k_gemm time %.2f ms' % (n, tot / 1e3))
for sh, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    tA, tB, M, N, K = sh
    print('%6.1f us total %4.1f%%  x%3d  tA %d tB %d  M %6d N %5d K %6d   %6.1f us each  %5.1f TF' % (t, 100 * t / tot, c, tA, tB, M, N, K, t / c, 2.0 * M * N * K * c / t / 1e6))
small = sum(v[1] for sh, v in agg.items() if max(sh[2], sh[4]) < 8192 or (sh[0] == 0 and sh[2] < 8192))
print('products with fewer than 8192 rows: %.2f ms' % (small / 1e3))
