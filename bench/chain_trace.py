#!/usr/bin/env python
"""The chain role of k_chol_df<true> from a DBAT_HIP_DF_TRACE dump (rows with i = -3, one per tile column; clocks of
df_chain_role in 10 ns ticks): per column how it was started (ticket / continued), how long it waited for T' and the
link, assembled, factored, looked at the flags, loaded, multiplied -- and the time from one factorisation's end to the next."""
import sys
import numpy as np

rows = np.loadtxt(sys.argv[1], delimiter=',', dtype=np.int64)
ch = rows[rows[:, 1] == -3]
T = ch[:, 3:19].astype(float) * 0.01
k = ch[:, 2]
t0 = rows[:, 3][rows[:, 3] > 0].min() * 0.01
used = T[:, 3] > 0
print('columns factored by the chain role: %d of %d' % (used.sum(), len(ch)))
byticket = ch[:, 3 + 8] == 1
pre = ch[:, 3 + 9]
print('T(p,k) requested under the last panel: %d' % (ch[:, 3 + 10] == 1).sum())
print('started by ticket %d, continued %d; T\' of the next column there at the first look %d, at the second %d'
      % ((used & byticket).sum(), (used & ~byticket).sum(), (pre == 1).sum(), (pre == 2).sum()))
names = ['wait', 'assemble', 'potf2', 'flags', 'loads', 'link', 'end']
d = np.diff(T[:, :8], axis=1)
for sel, what in ((used & ~byticket, 'continued'), (used & byticket, 'by ticket')):
    if sel.any():
        print('%-10s' % what + '  '.join('%s %.2f' % (n, np.median(d[sel, i])) for i, n in enumerate(names)) + '  (median us)')
# time between the ends of consecutive factorisations along the longest run of continued columns
order = np.argsort(k)
end3 = T[order, 3]
kk = k[order]
link = np.diff(end3)
ok = (np.diff(kk) == 1) & (end3[1:] > 0) & (end3[:-1] > 0)
cont = ~byticket[order][1:]
if (ok & cont).any():
    print('end of factorisation k -> end of k+1, continued links: median %.2f us, mean %.2f, n %d' % (np.median(link[ok & cont]), link[ok & cont].mean(), (ok & cont).sum()))
if (ok & ~cont).any():
    print('   ... links restarted by ticket: median %.2f us, n %d' % (np.median(link[ok & ~cont]), (ok & ~cont).sum()))
pf = rows[rows[:, 1] == -4]
if len(pf) == len(ch) and used.any() and (pf[:, 3 + 6] > 0).any():
    P = pf[:, 3:19].astype(float) * 0.01
    seq = np.stack([T[:, 2], P[:, 6], P[:, 7], P[:, 8], P[:, 9], P[:, 10], P[:, 11], P[:, 12], T[:, 3]], axis=1)
    dd = np.diff(seq, axis=1)[used]
    print('inside df_potf2 (median us): ' + '  '.join('%s %.2f' % (n, v) for n, v in zip(
        ['panel0', 'trail0', 'panel1', 'trail1', 'panel2', 'trail2', 'panel3', 'exit'], np.median(dd, axis=0))))
    el = (P[:, 13] - P[:, 5])[used & (P[:, 13] > 0)]
    if len(el):
        print('   panel 1: loads of the rows done -> elimination done (median us): %.2f' % np.median(el))
if len(sys.argv) > 2:
    for i in order:
        if T[i, 3] > 0:
            print('k %3d %s start %7.1f ' % (k[i], 'T' if byticket[i] else 'c', T[i, 0] - t0) + ' '.join('%5.2f' % x for x in d[i]) + '  pre %d' % pre[i])
