"""Percentiles of a kernel's launch durations from a rocprofv3 --kernel-trace CSV (the --stats summary only carries
min / average / max, and the max of ~10^4 launches is an outlier statistic).

    python profiles/tools/trace_percentiles.py <dir with *_kernel_trace.csv> <kernel-name substring> [...]
"""
import csv
import glob
import os
import sys

import numpy as np

d = sys.argv[1]
files = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
for sub in sys.argv[2:]:
    t = np.array([int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows if sub in r['Kernel_Name']]) / 1e3
    if not len(t):
        print('%s: no launches' % sub)
        continue
    print('%s: %d launches, us: min %.2f  p50 %.2f  mean %.2f  p90 %.2f  p99 %.2f  p99.9 %.2f  max %.2f  | launches above 28 us: %d '
          '(%.3f %%)' % (sub, len(t), t.min(), np.median(t), t.mean(), np.quantile(t, 0.9), np.quantile(t, 0.99),
                         np.quantile(t, 0.999), t.max(), (t > 28).sum(), 100 * (t > 28).mean()))
