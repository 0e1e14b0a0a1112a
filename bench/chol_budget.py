#!/usr/bin/env python
"""Where the workgroup-time of the dataflow Cholesky goes (DBAT_HIP_DF_TRACE dump, 100 MHz ticks):
python bench/chol_budget.py trace.csv n_workgroups n_products"""
import sys
import numpy as np

rows = np.loadtxt(sys.argv[1], delimiter=',', dtype=np.int64)
nwg, nprod = int(sys.argv[2]), int(sys.argv[3])
ti, tk, T = rows[:, 1], rows[:, 2], rows[:, 3:19].astype(float) * 0.01
T[:, 14:] = 0
t0 = T[T > 0].min()
T = np.where(T > 0, T - t0, np.nan)
span = np.nanmax(T[:, :5])
fact = ti >= 0
kinds = {'diagonal': fact & (ti == tk), 'off-diagonal': fact & (ti != tk), 'helpers (ti <= -2)': ti <= -2, 'backward': ti == -1}
tot = nwg * span
print('span %.1f us x %d workgroups = %.1f ms of workgroup time; %d products' % (span, nwg, tot * 1e-3, nprod))
acc = 0.0
for name, m in kinds.items():
    if not m.any():
        continue
    d = T[m, 4] - T[m, 0]
    s01 = np.nansum(T[m, 1] - T[m, 0])
    rest = np.nansum(d) - s01
    acc += np.nansum(d)
    print('%-20s %6d tasks: %.1f ms (%.1f %%): sums + waits %.1f ms, rest of the task %.1f ms (%.2f us per task)'
          % (name, m.sum(), np.nansum(d) * 1e-3, 100 * np.nansum(d) / tot, s01 * 1e-3, rest * 1e-3, rest / m.sum()))
print('between tasks (ticket, descriptor, idle at the end): %.1f ms (%.1f %%)' % ((tot - acc) * 1e-3, 100 * (tot - acc) / tot))
m = fact | (ti <= -2)
print('sums + waits of all product tasks: %.2f us per product' % (np.nansum(T[m, 1] - T[m, 0]) / nprod))
