#!/usr/bin/env python
"""Critical path of the dataflow Cholesky from a DBAT_HIP_DF_TRACE dump
(task, i, k, t[0..15] in 100 MHz ticks; chol_df.hpp).  Prints per-phase times of the
diagonal / off-diagonal / backward tasks and walks the chain of latest-finishing
dependencies from the last task back to the first."""
import sys
import numpy as np

rows = np.loadtxt(sys.argv[1], delimiter=',', dtype=np.int64)
cyc = (rows[:, 3 + 15] - rows[:, 3 + 14]).astype(float)      # s_memtime around df_potf2 (diagonal tasks)
rows[:, 3 + 14:] = 0
task, ti, tk, T = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3:19].astype(float) * 0.01   # us
t0 = T[T > 0].min()
T = np.where(T > 0, T - t0, np.nan)
fact = ti >= 0
diag = fact & (ti == tk)
off = fact & (ti != tk)
back = ti == -1                  # (ti <= -2: helpers of long sums)
print('tasks: %d diagonal, %d off-diagonal, %d backward; span %.1f us (factor %.1f, backward %.1f)'
      % (diag.sum(), off.sum(), back.sum(), np.nanmax(T), np.nanmax(T[fact, 4]), np.nanmax(T) - np.nanmax(T[fact, 4])))
ph = lambda m, a, b: np.nanmean(T[m, b] - T[m, a])
print('diagonal   : updates+waits %.2f | to potf2 %.2f | potf2 %.2f (panels %s) | stores %.2f | flag %.2f'
      % (ph(diag, 0, 1), ph(diag, 1, 2), ph(diag, 2, 12), ' '.join('%.2f' % ph(diag, 2 if q == 0 else 5 + q, 6 + q) for q in range(7)),
         ph(diag, 12, 3), ph(diag, 3, 4)))
print('off-diag   : updates+waits %.2f | wait diag %.2f | Linv load + product + stores %.2f | flag %.2f'
      % (ph(off, 0, 1), ph(off, 1, 2), ph(off, 2, 3), ph(off, 3, 4)))
dm = diag & (cyc > 0)
if dm.any():
    print('df_potf2   : %.0f s_memtime ticks = %.2f us -> %.2f GHz shader clock inside the kernel'
          % (cyc[dm].mean(), ph(dm, 2, 12), cyc[dm].mean() / ph(dm, 2, 12) * 1e-3))
if np.isfinite(T[diag, 5]).any():
    print('panel 1    : loads %.2f | elimination %.2f | stores + barrier %.2f us' % (ph(diag, 7, 5), ph(diag, 5, 13), ph(diag, 13, 8)))
print('backward   : %.2f us per panel' % ph(back, 0, 4))
# chain of diagonal completions
d_idx = np.flatnonzero(diag)
done = T[d_idx, 4]
order = np.argsort(tk[d_idx])
done = done[order]
gaps = np.diff(done)
print('diagonal tiles finish every %.2f us on average; %d gaps > 10 us (the dependent chain), their mean %.2f us'
      % (gaps.mean(), (gaps > 10).sum(), gaps[gaps > 10].mean() if (gaps > 10).any() else 0))
