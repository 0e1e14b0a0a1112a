"""Randomised sweep of the damping loops, outside the test-suite: small scenes of random shape and variant
(fixed IO, self-calibration, image-variant principal point, prior observations, four IO blocks), start values
pushed away from the solution by a random amount (so that Armijo halvings, rejected LM trials, dog-leg step types
and the failure codes all occur) -- bundle() on the device against the oracle's bundle() for GNA, LM, LMP and GM:
return code, iterate, sigma0, iteration history (tests/test_hip_parity.py::check_history), step lengths / step
types.  Prints one line per scene; exits non-zero at the end if anything disagreed.
    python bench/fuzz_solve.py [n_scenes] [first_seed]"""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, 'oracle')); sys.path.insert(0, os.path.join(R_, 'tests'))
import numpy as np
import dbat_oracle as o
from dbat_amd import bundle
from helpers import synth_struct, relerr
from test_hip_parity import check_history


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    nbad = 0
    codes = {}
    seeds = [int(v) for v in os.environ['FUZZ_SEEDS'].split(',')] if os.environ.get('FUZZ_SEEDS') else range(seed0, seed0 + n_scenes)
    for sd in seeds:
        rng = np.random.default_rng(9000 + sd)
        variant = str(rng.choice(['plain', 'selfcal', 'imagevar', 'priors', 'groups4']))
        name = str(rng.choice(['tiny', 'tiny', 'small']))
        s, truth = synth_struct(name, variant, seed=3000 + sd)
        push = float(rng.choice([0.0, 1.0, 3.0, 10.0, 30.0])) * float(os.environ.get('FUZZ_PUSH_SCALE', '1'))
        nc, npnt = s.EO.val.shape[1], s.OP.val.shape[1]
        s.EO.val[0:3] += push * rng.normal(0, 0.05, (3, nc))
        s.EO.val[3:6] += push * rng.normal(0, 0.002, (3, nc))
        s.OP.val += push * rng.normal(0, 0.05, (3, npnt))
        line = 'seed %3d: %-5s %-8s push %4.1f |' % (sd, name, variant, push)
        for damping in os.environ.get('FUZZ_DAMPINGS', 'gna,lm,lmp,gm').split(','):
            try:
                res, ok, iters, s0, E = bundle(s, damping)
                ro, oko, ito, s0o, Eo = o.bundle(s, damping)
                codes[Eo.code] = codes.get(Eo.code, 0) + 1
                assert ok == oko and E.code == Eo.code, 'code %d, oracle %d' % (E.code, Eo.code)
                if Eo.code == 0:
                    assert relerr(E.x, Eo.x) < 1e-6, 'x %.1e' % relerr(E.x, Eo.x)
                    assert abs(s0 - s0o) < 1e-7 * s0o, 'sigma0'
                if Eo.code in (0, -2):
                    try:
                        check_history(E, Eo, iters, ito, damping, s=None)
                    except AssertionError:
                        # far from the solution the iteration amplifies rounding differences step by step (the
                        # normal matrices there have condition numbers of 1e9 and more): same code, same number of
                        # iterations and a residual history that agrees to three digits is all that can be asked
                        k = min(len(E.res), len(Eo.res))
                        herr = float(np.max(np.abs(np.asarray(E.res[:k]) - np.asarray(Eo.res[:k])) / np.asarray(Eo.res[:k])))
                        assert iters == ito and herr < 1e-3, 'iterations %d / %d, residual history differs by %.1e' % (iters, ito, herr)
                        extra_note = ' (history to %.0e)' % herr
                    else:
                        extra_note = ''
                    if damping == 'gna' and not extra_note: assert np.array_equal(E.damping.alpha, Eo.damping.alpha), 'alpha'
                    if damping == 'lmp' and not extra_note: assert np.array_equal(E.damping.step, Eo.damping.step), 'step types'
                else:
                    extra_note = ''
                extra = ''
                if damping == 'gna' and Eo.code in (0, -2) and len(Eo.damping.alpha) and np.min(Eo.damping.alpha) < 1: extra = ' (alpha min %.3g)' % np.min(Eo.damping.alpha)
                line += ' %s %d/%d its code %d%s%s;' % (damping, iters, ito, Eo.code, extra, extra_note)
            except AssertionError as e:
                nbad += 1
                line += ' %s DISAGREES (%s);' % (damping, e)
        print(line, flush=True)
    print('%d scenes x 4 dampings; oracle return codes %s; %d disagreements' % (n_scenes, dict(sorted(codes.items())), nbad))
    sys.exit(1 if nbad else 0)


if __name__ == '__main__':
    main()
