import sys, time, numpy as np
sys.path.insert(0, '.')
from dbat_amd import synth, bundle, _hip
name = sys.argv[1]
t = time.time(); s, truth = synth.make_scene(name); print(name, 'gen %.1fs' % (time.time() - t), flush=True)
t = time.time()
res, ok, iters, s0, E = bundle(s, s.damping, store_trace=False)
print(name, s.damping, 'ok', ok, 'iters', iters, 'code', E.code, 's0 %.6f' % s0, 'solve time %.2fs' % E.time, 'total %.1fs' % (time.time() - t), 'counters', E.counters, flush=True)
print(' OP err std', np.abs(res.OP.val - truth['OP']).std(), ' res', E.res[:8])
if name in ('C2', 'C4'):
    print(' IO est', res.IO.val[[0, 1, 2, 5, 6, 7, 8, 9], 0], ' truth', truth['IO'][[0, 1, 2, 5, 6, 7, 8, 9], 0])
