import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
seq = sorted(((r['Kernel_Name'].split('(')[0].replace('void ', '').replace('qtos::', ''), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows), key=lambda x: x[1])
starts = [i for i, s in enumerate(seq) if s[0].startswith('k_start')]
i0, i1 = starts[-2], starts[-1]
prev = None
for name, st, en in seq[i0:i1]:
    print("%-26s %8.1f us  gap %6.1f" % (name[:26], (en - st) / 1e3, (st - prev) / 1e3 if prev else 0)); prev = en
print("solve total %.1f us" % ((seq[i1][1] - seq[i0][1]) / 1e3))
