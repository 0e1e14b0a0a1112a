"""Repeatability of linearise+solve on one GPU (races show up as run-to-run differences
far above rounding)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dbat_amd import synth, _hip
for name, reps in (('small', 30), ('C1', 30), ('C3', 5)):
    s, _ = synth.make_scene(name)
    h = _hip.Handle(s)
    x0 = h.serialize()
    ref = None; worst = 0.0
    for i in range(reps):
        p, st = h.linearize_solve(x0, 0.0, True)
        if ref is None: ref = p.copy()
        worst = max(worst, np.linalg.norm(p - ref) / np.linalg.norm(ref))
    print(name, 'max run-to-run rel. difference of the step over', reps, 'runs:', worst, 'singular', st['singular'])
    h.close()
