import csv, glob, sys, re
f = glob.glob(sys.argv[1] + '/*/*memory_copy_trace.csv')[0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction']) for r in csv.DictReader(open(f)))
calls = [[ev[0]]]
for e in ev[1:]:
    if e[0] - calls[-1][-1][1] > 30e6: calls.append([])
    calls[-1].append(e)
c = calls[-1]; t0 = c[0][0]
d2 = [(s, e) for s, e, d in c if 'DEVICE_TO_DEVICE' in d]
tiles = [tuple(map(int, m.groups())) for m in re.finditer(r"\[mxgpu\] tile (\d+) (\d+)", open(sys.argv[2]).read())]
tiles = tiles[-len(d2):]
# last block tiles have up to 3 copies; ignore: just print first 140
k = 0
out = []
for (b, g) in tiles:
    if k >= len(d2): break
    s, e = d2[k]; out.append((b, g, round((s - t0) / 1e6, 1), round((e - s) / 1e6, 2))); k += 1
    if b == 15 and g < 15: k += 2
print(out[:24]); print(out[80:130])
slow = [(b, g) for b, g, t, d in out if d > 1.0]
import collections
print("slow by block", sorted(collections.Counter(b for b, g in slow).items()))
print("slow by piece", sorted(collections.Counter(g for b, g in slow).items()))
