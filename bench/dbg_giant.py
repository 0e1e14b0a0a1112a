import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
os.environ['DBAT_HIP_BT'] = os.environ.get('BTX', '128')
os.environ['DBAT_HIP_GIANT_THREADS'] = os.environ.get('GT', '64')
import numpy as np, scipy.sparse as sp
import dbat_oracle as o
from dbat_amd import synth, _hip
from helpers import relerr
s, truth = synth.make_scene('small', cams=140, points=500, rays=6)
s.IO.val[5:10] = 0.0; truth['IO'][5:10] = 0.0
nc = s.EO.val.shape[1]; px = float(np.ravel(s.IO.sensor.pxSize)[0]); rng = np.random.default_rng(5)
add_cam, add_pt = [], []
for p in (3, 77, 250):
    have = set(s.IP.cam[s.IP.pt == p].tolist())
    for c in range(nc):
        if c not in have: add_cam.append(c); add_pt.append(p)
cam = np.r_[s.IP.cam, np.array(add_cam)]; pt = np.r_[s.IP.pt, np.array(add_pt)]
order = np.lexsort((pt, cam)); cam, pt = cam[order], pt[order]
uv, depth = synth.project(truth['IO'], truth['EO'], truth['OP'], cam, pt, px)
s.IP.val = uv + rng.normal(0, 0.5, uv.shape); s.IP.std = np.ones_like(uv); s.IP.cam, s.IP.pt = cam, pt
so = o.buildserialindices(s); x0 = o.serialize(so); w = o.buildweightvector(so)
R = np.sqrt(w)
r_o, K = o.brown_euler_cam4(x0, so, jac=True)
J = (sp.diags(R) @ K).tocsc()
p_o, sing, *_ = o._scaled_gn(J, R * r_o)
h = _hip.Handle(s)
print(h.info())
p_h, st = h.linearize_solve(x0, 0.0, True)
print('oracle singular', sing, 'hip', st)
print('relerr step', relerr(p_h, p_o))
print('grad relerr', relerr(h.gradient(), J.T @ (R * r_o)))
print('colnorms relerr', relerr(h.colnorms(), np.sqrt(np.asarray(J.multiply(J).sum(0)).ravel())))
