#!/usr/bin/env python3
"""Timeline of the LAST export call in a `rocprofv3 --memory-copy-trace --output-format csv -d DIR` run: per 20 ms, how
many downloads (the runtime labels copies into registered host memory DEVICE_TO_DEVICE) started, their mean duration and
the time the download queue was busy; the uploads' busy time beside it.   python tools/copy_timeline.py DIR"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*memory_copy_trace.csv')[0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction']) for r in csv.DictReader(open(f)))
calls = [[ev[0]]]
for e in ev[1:]:
    if e[0] - calls[-1][-1][1] > 30e6:
        calls.append([])
    calls[-1].append(e)
c = calls[-1]
t0 = c[0][0]
d2 = [(s, e) for s, e, d in c if 'DEVICE_TO_DEVICE' in d or 'DEVICE_TO_HOST' in d]
h2 = [(s, e) for s, e, d in c if 'HOST_TO_DEVICE' in d]
print("downloads: %d, first at %.1f ms, last ends %.1f ms; uploads end %.1f ms" % (len(d2), (d2[0][0] - t0) / 1e6, (d2[-1][1] - t0) / 1e6, (h2[-1][1] - t0) / 1e6))
b = collections.defaultdict(list)
for s, e in d2:
    b[int((s - t0) / 20e6)].append((e - s) / 1e6)
bh = collections.defaultdict(float)
for s, e in h2:
    bh[int((s - t0) / 20e6)] += (e - s) / 1e6
for k in sorted(set(b) | set(bh)):
    v = b.get(k, [])
    print("%4d ms: downloads n=%3d mean %.3f ms busy %5.1f | uploads busy %5.1f" % (k * 20, len(v), sum(v) / len(v) if v else 0, sum(v), bh.get(k, 0)))
