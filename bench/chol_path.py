#!/usr/bin/env python
"""Critical path of one k_chol_df launch from a DBAT_HIP_DF_TRACE dump: starting from the task that
finishes last, follow the dependency that arrived last, and say for every link how long the task
worked after that arrival (serial tail) -- or, if it was taken from the queue AFTER its last input
existed, how long the input waited for a workgroup (queueing)."""
import sys
import numpy as np

rows = np.loadtxt(sys.argv[1], delimiter=',', dtype=np.int64)
ti, tk = rows[:, 1], rows[:, 2]
T = rows[:, 3:19].astype(float) * 0.01
jlo, jhi = rows[:, 19], rows[:, 20]                 # the task's own products: j in [jlo, jhi)
mode = rows[:, 21] if rows.shape[1] > 21 else np.zeros(len(rows), int)   # 1: sum only (T published), 2: diagonal task that finishes that tile
helper = ti <= -2                                   # pieces of long sums: i = -(ti + 2)
t0 = T[:, 0][T[:, 0] > 0].min()
fact = (ti >= 0) | helper
ri = np.where(helper, -(ti + 2), ti)                # tile row of every factor task
idx = {(int(i), int(k)): n for n, (i, k) in enumerate(zip(ti, tk)) if i >= 0}      # tile -> the task that PUBLISHES it
sumtask = {}
for n in np.flatnonzero((ti >= 0) & (mode == 1)):
    sumtask[int(tk[n])] = n
    idx[(int(ti[n]), int(tk[n]))] = None           # filled below: published by the diagonal task of the column
for key in [kk for kk, v in idx.items() if v is None]:
    idx[key] = idx[(key[1], key[1])]
helpers = {}
for n in np.flatnonzero(helper):
    helpers.setdefault((int(ri[n]), int(tk[n])), []).append(n)
start = T[:, 0] - t0
sumdone = T[:, 1] - t0
done = T[:, 4] - t0
cols = {}
for (i, k) in idx:
    cols.setdefault(i, set()).add(k)
nprod = np.zeros(len(rows), int)
def own_js(n):
    i, k = int(ri[n]), int(tk[n])
    return [j for j in cols[k] if jlo[n] <= j < jhi[n] and j in cols[i]]
for n in np.flatnonzero(fact):
    nprod[n] = len(own_js(n))
print('tasks %d (%d helpers), products %d, span of the factorisation %.0f us' % (fact.sum(), helper.sum(), nprod.sum(), done[fact].max()))
q = np.quantile(nprod[fact], [0.5, 0.9, 0.99, 1.0])
print('products per task: median %d, 90%% %d, 99%% %d, max %d; tasks with > 64 products: %d holding %.0f%% of the products'
      % (q[0], q[1], q[2], q[3], (nprod > 64).sum(), 100.0 * nprod[nprod > 64].sum() / max(nprod.sum(), 1)))
hv = fact & (nprod >= 32)
if hv.any():
    per = (sumdone[hv] - start[hv]) / nprod[hv]
    print('tasks with >= 32 products: time in the sum per product: min %.2f, 10%% %.2f, median %.2f us (waiting included)'
          % (per.min(), np.quantile(per, 0.1), np.median(per)))
# walk back
n = int(np.argmax(np.where(fact, done, -1)))
chain = []
while True:
    i, k = int(ri[n]), int(tk[n])
    js = own_js(n)
    deps = [idx[(k, j)] for j in js] + [idx[(i, j)] for j in js if i != k]
    if not helper[n]:
        deps += helpers.get((i, k), [])
        if i != k and mode[n] == 0: deps.append(idx[(k, k)])
        if mode[n] == 2: deps.append(sumtask[k])
    if not deps:
        chain.append((int(ti[n]), k, nprod[n], start[n], 0.0, done[n] - start[n], 0.0)); break
    d = max(deps, key=lambda m: done[m])
    arrive = done[d]
    queue = max(start[n] - arrive, 0.0)          # the input existed before a workgroup took the task
    tail = done[n] - max(arrive, start[n])
    chain.append((int(ti[n]), k, nprod[n], start[n], arrive, tail, queue))
    n = d
chain.reverse()
tot_tail = sum(c[5] for c in chain); tot_q = sum(c[6] for c in chain)
print('critical path: %d tasks; work after the last input %.0f us, waiting for a workgroup %.0f us' % (len(chain), tot_tail, tot_q))
nd = sum(1 for c in chain if c[0] == c[1])
print('  %d diagonal tasks (mean tail %.1f us), %d off-diagonal (mean tail %.1f us)'
      % (nd, np.mean([c[5] for c in chain if c[0] == c[1]] or [0]), len(chain) - nd, np.mean([c[5] for c in chain if c[0] != c[1]] or [0])))
big = sorted(chain, key=lambda c: -(c[5] + c[6]))[:12]
print('  largest links (i, k, products, start, last input, tail, queued):')
for c in big:
    print('   (%3d,%3d) %4d products  start %7.1f  input %7.1f  tail %6.1f  queued %6.1f' % c)
