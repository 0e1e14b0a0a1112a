#!/usr/bin/env python
"""Per-task clocks of a DBAT_HIP_DF_TRACE dump: how long the sums take per product, by length of the sum, and where the
workgroups' time goes (sum / rest of the task / between tasks)."""
import sys
import numpy as np

rows = np.loadtxt(sys.argv[1], delimiter=',', dtype=np.int64)
T = rows[:, 3:19].astype(float) * 0.01
i, k = rows[:, 1], rows[:, 2]
jlo, jhi = rows[:, 19], rows[:, 20]
task = (i >= 0) | (i <= -2)
task &= T[:, 0] > 0
t0 = T[task, 0].min()
span = T[task, 4].max() - t0
sum_t = T[task, 1] - T[task, 0]
rest = T[task, 4] - T[task, 1]
print('tasks %d, span %.0f us; time in the sums %.0f us (%.2f of the workgroup time at 256), rest of the tasks %.0f us' % (
    task.sum(), span, sum_t.sum(), sum_t.sum() / (256 * span), rest.sum()))
width = (jhi - jlo)[task]
for lo, hi in ((0, 1), (1, 4), (4, 12), (12, 40), (40, 10 ** 9)):
    m = (width >= lo) & (width < hi)
    if m.any():
        print('  column range [%d, %d): %6d tasks, sum time mean %.2f median %.2f us, total %.0f us' % (lo, hi, m.sum(), sum_t[m].mean(), np.median(sum_t[m]), sum_t[m].sum()))
