"""Gap statistics of the last `n_ms` milliseconds of a rocprofv3 kernel trace."""
import collections, csv, glob, sys
d, win_ms = sys.argv[1], float(sys.argv[2])
kt = list(csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0])))
kt.sort(key=lambda r: int(r['Start_Timestamp']))
if win_ms > 0:
    t_hi = int(kt[-1]['End_Timestamp'])
    K = [r for r in kt if int(r['Start_Timestamp']) >= t_hi - win_ms * 1e6]
else:
    # exactly one training iteration: between the ends of the last two optimizer-step clusters (multi_tensor_apply)
    ends, last = [], None
    for r in kt:
        if 'multi_tensor_apply' in r['Kernel_Name']:
            t = int(r['End_Timestamp'])
            if last is None or t - last > 50e6: ends.append(t)
            else: ends[-1] = t
            last = t
    lo, hi = ends[-2], ends[-1]
    K = [r for r in kt if lo < int(r['Start_Timestamp']) and int(r['End_Timestamp']) <= hi]
    win_ms = (hi - lo) / 1e6
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in K) / 1e6
print("window %.1f ms, %d kernels, kernel time %.2f ms" % (win_ms, len(K), busy))
gaps = [((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3, a['Kernel_Name'][:45], b['Kernel_Name'][:45], b['Queue_Id'], b['Stream_Id']) for a, b in zip(K[:-1], K[1:])]
for th in (0.5, 1, 2, 5, 10, 20, 50, 100, 500):
    sel = [g for g in gaps if g[0] > th]
    print("gaps > %5.1f us: %5d, %.2f ms" % (th, len(sel), sum(g[0] for g in sel) / 1e3))
print("overlaps (negative gaps):", sum(1 for g in gaps if g[0] < 0), "queues:", collections.Counter(g[3] for g in gaps), "streams:", collections.Counter(g[4] for g in gaps))
for g in sorted(gaps, reverse=True)[:15]:
    print("  %8.1f  %s -> %s" % g[:3])
tot = collections.Counter(); cnt = collections.Counter()
for r in K:
    k = r['Kernel_Name'][:70]; tot[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6; cnt[k] += 1
print("per kernel (ms, calls):")
for k, v in tot.most_common(40):
    print("  %8.3f %6d  %s" % (v, cnt[k], k))
