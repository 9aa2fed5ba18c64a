"""Developer: per-search kernel timeline from a rocprofv3 --kernel-trace CSV: kernel durations and the gaps between them (us), averaged
over the steady-state searches. usage: python tools/gap_trace.py <kernel_trace.csv> <first kernel name substring>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2]
seqs, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    if first in n:
        if cur: seqs.append(cur)
        cur = []
    if cur is not None and ("rdx" in n):
        cur.append((n[:48], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
seqs = [s for s in seqs[5:-1] if len(s) == len(seqs[len(seqs) // 2])]
acc = collections.OrderedDict()
for s in seqs:
    for i, (n, a, b) in enumerate(s):
        d = acc.setdefault((i, n), [0.0, 0.0, 0])
        d[0] += (b - a) / 1e3
        d[1] += ((a - s[i - 1][2]) / 1e3) if i else 0.0
        d[2] += 1
tot = 0
for (i, n), (dur, gap, c) in acc.items():
    print(f"{i} {n:50s} gap before {gap / c:6.2f} us   kernel {dur / c:7.2f} us")
    tot += (dur + gap) / c
print("first start -> last end: %.2f us over %d searches; start-to-start %.2f us" % (tot, len(seqs), (seqs[-1][0][1] - seqs[0][0][1]) / 1e3 / max(1, len(seqs) - 1)))
