"""Reads a rocprofv3 --hip-trace --kernel-trace CSV pair and says, for the last part of the run, how far ahead of the GPU
the host was at every kernel start, which HIP calls blocked, and what the host was doing during the big GPU gaps."""
import collections, csv, glob, sys
d = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0])))
at = list(csv.DictReader(open(glob.glob(d + '/**/*hip_api_trace.csv', recursive=True)[0])))
print("kernels", len(kt), "api calls", len(at), "api columns", list(at[0].keys()))
kt.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8792
K = kt[-n:]
t_lo = int(K[0]['Start_Timestamp'])
api_by_corr = {r['Correlation_Id']: r for r in at}
names = collections.Counter(r['Function'] for r in at if int(r['Start_Timestamp']) >= t_lo)
print("api calls in window:", names.most_common(14))
# blocking calls
blk = collections.Counter(); blkn = collections.Counter()
for r in at:
    if int(r['Start_Timestamp']) < t_lo: continue
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if dur > 30:
        blk[r['Function']] += dur; blkn[r['Function']] += 1
print("calls > 30 us (total ms, count):", [(k, round(v / 1e3, 2), blkn[k]) for k, v in blk.most_common(10)])
leads = []
for a, b in zip(K[:-1], K[1:]):
    api = api_by_corr.get(b['Correlation_Id'])
    if api is None: continue
    gap = (int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3
    lead = (int(a['End_Timestamp']) - int(api['End_Timestamp'])) / 1e3   # > 0: launch was queued before the previous kernel ended
    leads.append((gap, lead, a['Kernel_Name'][:40], b['Kernel_Name'][:40]))
import statistics
L = [l for _, l, _, _ in leads]
print("matched", len(leads), "median lead us", statistics.median(L), "share with lead < 0:", sum(1 for l in L if l < 0) / len(L))
g_late = sum(g for g, l, _, _ in leads if g > 1 and l < 0); g_early = sum(g for g, l, _, _ in leads if g > 1 and l >= 0)
print("gap time with the launch issued AFTER the previous kernel ended (host late): %.2f ms; with launch already queued: %.2f ms" % (g_late / 1e3, g_early / 1e3))
print("largest gaps where the launch was already queued:")
for g, l, a, b in sorted([x for x in leads if x[1] >= 0], reverse=True)[:12]:
    print("  gap %7.1f lead %9.1f  %s -> %s" % (g, l, a, b))
# timeline of lead in 10 ms bins
bins = collections.defaultdict(list)
for (g, l, a, b), kb in zip(leads, K[1:]):
    bins[(int(kb['Start_Timestamp']) - t_lo) // 10_000_000].append(l)
print("median lead (us) per 10 ms:", [(k, round(statistics.median(v))) for k, v in sorted(bins.items())])
