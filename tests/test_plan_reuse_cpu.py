"""Plan reuse, host side (no GPU): the structure key (dbat_hip_structure_key) is blind to the parameter and prior VALUES and
sees everything a plan depends on -- masks, blocks, visibility, observations, prior pattern, shard, device, DBAT_HIP_*
switches (bundle.m:156-175 rebuilds the serial indices only when they are missing; deserialize.m:31-46)."""
import copy

import numpy as np
import pytest

from helpers import synth_struct


def test_structure_key_ignores_values_and_sees_structure(monkeypatch):
    from dbat_amd import _hip
    s, _ = synth_struct('tiny', 'priors')
    k0 = _hip.structure_key(s)
    assert k0 == _hip.structure_key(copy.deepcopy(s))
    # values: not part of the key
    t = copy.deepcopy(s)
    t.EO.val = t.EO.val + 0.01; t.OP.val = t.OP.val * 1.001; t.IO.val = t.IO.val * 1.01
    t.prior.EO.val = t.prior.EO.val + 0.5; t.prior.OP.std = t.prior.OP.std * 2
    assert _hip.structure_key(t) == k0
    # structure: every one of these is
    def changed(f):
        t = copy.deepcopy(s)
        f(t)
        return _hip.structure_key(t) != k0
    assert changed(lambda t: t.bundle.est.OP.__setitem__((0, 7), False))           # a mask
    assert changed(lambda t: t.bundle.est.EO.__setitem__((3, 2), False))
    assert changed(lambda t: t.bundle.est.IO.__setitem__((0, slice(None)), True))
    assert changed(lambda t: t.IO.struct.block.__setitem__((1, 0), 9))              # a block
    assert changed(lambda t: t.prior.OP.use.__setitem__((slice(None), 1), True))     # the prior pattern
    assert changed(lambda t: t.IP.val.__setitem__((0, 5), t.IP.val[0, 5] + 1e-9))    # an image observation
    assert changed(lambda t: t.IP.std.__setitem__((1, 5), 2.5))
    assert changed(lambda t: t.IO.sensor.pxSize.__setitem__((0, 0), 1.0))
    def drop(t):                                                                     # visibility
        keep = np.ones(t.IP.cam.size, bool); keep[3] = False
        t.IP.cam, t.IP.pt, t.IP.val, t.IP.std = t.IP.cam[keep], t.IP.pt[keep], t.IP.val[:, keep], t.IP.std[:, keep]
    assert changed(drop)
    assert _hip.structure_key(s, shard_rank=1, shard_count=2) != k0
    assert _hip.structure_key(s, device=1) != k0
    monkeypatch.setenv('DBAT_HIP_CMAX', '6')                                         # a switch of the library
    assert _hip.structure_key(s) != k0


def test_structure_key_is_the_same_for_any_thread_count(monkeypatch):
    from dbat_amd import _hip, synth
    s, _ = synth.make_scene('C1')                                                    # 3 MB of observations: several blocks
    # (DBAT_HIP_PLAN_THREADS is itself a DBAT_HIP_* variable and so part of the key: equal structs must give equal keys
    # under every thread count, and the blocks of a field are combined in order whatever thread hashed them)
    for nt in ('1', '8'):
        monkeypatch.setenv('DBAT_HIP_PLAN_THREADS', nt)
        a = _hip.structure_key(s); b = _hip.structure_key(copy.deepcopy(s))
        assert a == b
        monkeypatch.delenv('DBAT_HIP_PLAN_THREADS')


def test_marshalling_a_struct_makes_no_copy_of_the_observations():
    """problem_from_struct hands the library views of IP.* (Fortran-ordered float64 / int32 arrays as dbat_amd.dbatstruct
    builds them): the second bundle() on a 10 M observation project must not pay a pass over 500 MB to find its handle."""
    from dbat_amd import _hip, synth
    s, _ = synth.make_scene('small')
    p, keep = _hip.problem_from_struct(s)
    for name, arr in (('ip_val', s.IP.val), ('ip_std', s.IP.std), ('ip_cam', s.IP.cam), ('ip_pt', s.IP.pt), ('OP_val', s.OP.val)):
        assert np.shares_memory(keep[name], arr), name
