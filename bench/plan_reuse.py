"""Plan reuse at bench size: wall time of bundle() on a scene, of a second bundle() on the same
structure (the cached handle: structure key + dbat_hip_set_values), and of bundle_cov on that handle.
    python bench/plan_reuse.py [C3]"""
import json, os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
import numpy as np
import torch  # noqa: F401
from dbat_amd import _hip, bundle, bundle_cov, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
s, _ = synth.make_scene(cfg)
damping = 'lmp' if cfg == 'C1' else 'lm'
h0 = _hip.Handle(synth.make_scene('tiny')[0]); h0.close()          # library load, HIP context: once per process
_hip.clear_cache()
out = {'config': cfg, 'runs': []}
for i in range(3):
    t = time.perf_counter()
    r = bundle(s, damping, store_trace=False)
    wall = time.perf_counter() - t
    E = r[4]
    out['runs'].append({'wall_s': wall, 'solve_time_s': E.time, 'iters': r[2], 'sigma0': r[3], **E.timeHost})
t = time.perf_counter()
p, keep = _hip.problem_from_struct(s)
t1 = time.perf_counter()
k = _hip.structure_key(None, _problem=(p, keep))
t2 = time.perf_counter()
out['marshal_s'] = t1 - t; out['structure_key_s'] = t2 - t1
if cfg in ('C1', 'C2', 'C3'):
    t = time.perf_counter()
    C = bundle_cov(r[0], r[4], 'CEO')
    out['bundle_cov_s'] = time.perf_counter() - t
out['cache'] = dict(_hip.cache_stats)
print(json.dumps(out))
