#!/usr/bin/env python
"""One compact line per bench.py run (measurement helper): python bench/quick.py C3 [ENV=VALUE ...] [-- extra bench args]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if '--' in args:
    i = args.index('--')
    args, extra = args[:i], args[i + 1:]
cfg = args[0]
env = dict(os.environ)
for kv in args[1:]:
    k, v = kv.split('=', 1)
    env[k] = v
p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', cfg, '--no-cpu-baseline', '--no-solve', '--steps', '10'] + extra,
                   env=env, capture_output=True, text=True)
tag = ' '.join(args[1:]) or 'default'
err = [l for l in p.stderr.splitlines() if l.startswith('[')]
try:
    d = json.loads(p.stdout.strip().splitlines()[-1])
    km = d['kernel_ms']
    print('%s %-40s step %.3f build %.3f chol %.3f backsub %.3f resid %.3f | tile kernel %.4f chol %.4f bs %.4f res %.4f | tiles %d' % (
        cfg, tag, d['ms_per_step'], d['ms_build_schur'], d['ms_factor_solve'], d['ms_backsub'], d['ms_trial_residual'],
        list(km.values())[0], km['k_chol_df'], km['k_backsub'], km['k_residual_cm'], d['config']['n_tiles']))
except Exception as e:
    print(cfg, tag, 'FAILED', e, p.stderr[-2000:])
for l in err[-3:]:
    print('   ', l)
