"""Deterministic mode: repeats bit for bit?  and what does it cost?  python bench/r05_det.py C1 C2 C3 [small ...]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from dbat_amd import _hip, synth
for name in sys.argv[1:]:
    sel = name.endswith('+io')
    base = name[:-3] if sel else name
    s, _ = synth.make_scene(base, selfcal=True) if sel else synth.make_scene(base)
    h = _hip.Handle(s)
    x0 = h.serialize()
    p_def, st = h.linearize_solve(x0, 0.0, True)
    t0 = time.perf_counter()
    for _ in range(5): h.linearize_solve(x0, 0.0, True)
    t_def = (time.perf_counter() - t0) / 5
    try:
        h.set_deterministic(True)
    except _hip.DbatHipError as e:
        print(name, 'deterministic mode refused:', e); h.close(); continue
    ref = None; bad = 0
    t0 = time.perf_counter()
    for i in range(10):
        p, st = h.linearize_solve(x0, 0.0, True)
        if ref is None: ref = p.copy()
        nb = int(np.sum(p != ref))
        bad += nb > 0
        if nb: print('   repeat', i, 'differs in', nb, 'of', len(p), 'entries, rel', np.linalg.norm(p - ref) / np.linalg.norm(ref))
    t_det = (time.perf_counter() - t0) / 10
    print('%-8s deterministic: %d of 10 repeats differ; vs default %.2e; singular %s; linearise+solve %.3f ms (default %.3f ms)'
          % (name, bad, np.linalg.norm(ref - p_def) / np.linalg.norm(p_def), st['singular'], t_det * 1e3, t_def * 1e3), flush=True)
    h.close()
