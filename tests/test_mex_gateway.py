"""The MATLAB side of the boundary, checked without MATLAB (CPU only).

(a) mex/dbat_hip_mex.cpp passes the compiler's front end against a declarations-only mex.h
    (tests/mex_stub/mex.h; prototypes as used by the reference's own MEX file,
    code/test/postcov/icpc_mex.c:495-611) and the real include/dbat_hip.h;
(b) the field names the gateway reads (field(P,"...") / scalar(O,"...")) are exactly the names of
    the struct literals in matlab/bundle_hip.m, and the output list of the call in bundle_hip.m
    matches the gateway's documented [x,code,...] order and its highest plhs index.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GATEWAY = os.path.join(ROOT, 'mex', 'dbat_hip_mex.cpp')
WRAPPER = os.path.join(ROOT, 'matlab', 'bundle_hip.m')


def test_gateway_passes_the_compiler_front_end():
    cxx = shutil.which('g++') or shutil.which('c++')
    if cxx is None:
        pytest.skip('no C++ compiler')
    r = subprocess.run([cxx, '-std=c++17', '-fsyntax-only', '-Wall', '-Wextra', '-Werror',
                        '-I' + os.path.join(ROOT, 'tests', 'mex_stub'), '-I' + os.path.join(ROOT, 'include'), GATEWAY],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _matlab_struct_fields(src, var):
    """Field names of `var=struct('a',...,'b',...)` (continuation lines joined)."""
    m = re.search(r"^%s=struct\((.*?)\);" % re.escape(var), src.replace('...\n', ' '), re.M | re.S)
    assert m, 'no struct literal for %s' % var
    body = m.group(1)
    # top-level 'name', value pairs: a name is a quoted identifier directly followed by a comma at depth 0
    names, depth, i = [], 0, 0
    expect_name = True
    while i < len(body):
        ch = body[i]
        if ch in '([{':
            depth += 1
        elif ch in ')]}':
            depth -= 1
        elif ch == ',' and depth == 0:
            expect_name = not expect_name
        elif ch == "'" and depth == 0 and expect_name:
            j = body.index("'", i + 1)
            names.append(body[i + 1:j])
            i = j
        elif ch == "'":                      # a string inside a value: skip it
            i = body.index("'", i + 1)
        i += 1
    return names


def test_field_names_and_output_order_agree():
    cpp = open(GATEWAY).read()
    m = open(WRAPPER).read()
    read_P = set(re.findall(r'(?:field|scalar)\(P,\s*"(\w+)"\)', cpp))
    read_O = set(re.findall(r'(?:field|scalar)\(O,\s*"(\w+)"\)', cpp))
    P_fields = _matlab_struct_fields(m, 'P')
    O_fields = _matlab_struct_fields(m, 'opt')
    assert len(P_fields) == len(set(P_fields)) and len(O_fields) == len(set(O_fields))
    assert read_P == set(P_fields), (sorted(read_P - set(P_fields)), sorted(set(P_fields) - read_P))
    assert read_O == set(O_fields), (sorted(read_O - set(O_fields)), sorted(set(O_fields) - read_O))
    # outputs: the call in bundle_hip.m, the gateway's documented list, and the plhs indices it fills
    call = re.search(r'^\[([^\]]+)\]=dbat_hip_mex\(P,opt\);', m, re.M)
    assert call
    outs_m = [v.strip() for v in call.group(1).split(',')]
    doc = re.search(r'//\s*\[([^\]]+)\]\s*=\s*dbat_hip_mex\(P, opt\)', cpp)
    assert doc
    outs_cpp = [v.strip() for v in doc.group(1).split(',')]
    alias = {'sigma0': 's0', 'CEO': 'CEOb', 'CIO': 'CIOu', 'COP': 'COPb'}
    assert [alias.get(v, v) for v in outs_cpp] == outs_m
    plhs = sorted({int(i) for i in re.findall(r'plhs\[(\d+)\]\s*=', cpp)})
    assert plhs == list(range(len(outs_m))), (plhs, outs_m)
    # every output beyond the first is guarded by the matching nlhs test (MATLAB always accepts plhs[0])
    for i in plhs[1:]:
        guards = [int(g) for g in re.findall(r'if \(nlhs > (\d+)\)[^\n;]*plhs\[%d\]\s*=' % i, cpp)]
        outer = [int(g) for g in re.findall(r'if \(nlhs > (\d+)\) \{', cpp)]
        assert any(g == i for g in guards) or any(g <= i for g in outer), i   # nlhs > i <=> plhs[i] may be set


def _option_cases(src):
    """{option string: statement} of the varargin switch of a bundle()-style parser."""
    sw = src[src.index('switch lower('):]
    sw = sw[:sw.index('\n        end')]
    out = {}
    for m in re.finditer(r"^\s*case\s+(\{[^}]*\}|'[^']*')\s*,?\s*(.*)$", sw, re.M):
        for name in re.findall(r"'([^']*)'", m.group(1)):
            out[name] = m.group(2).strip()
    return out


def test_every_option_bundle_accepts_reaches_the_gateway_or_the_result():
    """bundle.m:78-132 accepts a fixed set of option strings.  bundle_hip.m must accept the same set, and none may be
    parsed and dropped (round 4's wrapper swallowed 'pmdof' / 'dofverb'): the variable a case sets has to be read
    again -- in the gateway's opt struct or in the result code after the call."""
    m = open(WRAPPER).read()
    cases = _option_cases(m)
    ref_opts = {'none', 'gm', 'gna', 'lm', 'lmp', 'trace', 'singulartest', 'nosingulartest', 'pmdof', 'dofverb', 'absterm'}
    ref = '/root/reference/code/bundle/bundle.m'
    if os.path.exists(ref):                     # (not on the GPU box; the set above is bundle.m:97-121)
        assert set(_option_cases(open(ref).read())) == ref_opts
    assert set(cases) == ref_opts, (sorted(set(cases) ^ ref_opts))
    body_after_parse = m[m.index("% --- what BUNDLE has no argument for"):]
    opt_literal = re.search(r"^opt=struct\((.*?)\);", m.replace('...\n', ' '), re.M | re.S).group(1)
    for name, stmt in cases.items():
        assert stmt and not stmt.startswith('%'), 'option %r is parsed and dropped' % name
        var = re.match(r"(\w+)\s*=", stmt)
        assert var, (name, stmt)
        v = var.group(1)
        uses = len(re.findall(r'\b%s\b' % v, body_after_parse))
        assert uses >= 1, 'option %r sets %s, which nothing reads' % (name, v)
    # what must reach the gateway does
    for v in ('maxIter', 'convTol', 'absTerm', 'singularTest', 'dampNo'):
        assert re.search(r'\b%s\b' % v, opt_literal), v
    # what acts after the call does: the degrees of freedom and their report
    tail = m[m.index('=dbat_hip_mex(P,opt);'):]
    assert re.search(r'if pmDof\b', tail) and re.search(r'dof=lenR\+p-lenX', tail) and re.search(r'E\.redundancy=dof', tail)
    assert re.search(r'if dofVerb\b', tail) and re.search(r'if doTrace\b', tail)
    assert 's0=sqrt((rw\'*rw)/dof)' in tail


def test_gpu_side_fields_reach_the_gateway():
    """s.bundle.hip.* (device, shardRank, shardCount, commId, wantJ, wantCov, deterministic): every default field of
    the wrapper is copied into opt, an unknown field raises, and the gateway reads each of them (the set equality of
    test_field_names_and_output_order_agree) -- round 4's gateway hard-coded device 0 / one rank."""
    m = open(WRAPPER).read()
    cpp = open(GATEWAY).read()
    hip_fields = _matlab_struct_fields(m, 'hip')
    assert set(hip_fields) == {'device', 'shardRank', 'shardCount', 'commId', 'wantJ', 'wantCov', 'deterministic',
                               'termFun', 'vetoFun'}
    opt_literal = re.search(r"^opt=struct\((.*?)\);", m.replace('...\n', ' '), re.M | re.S).group(1)
    for f in hip_fields:
        assert re.search(r'hip\.%s\b' % f, opt_literal) or (f == 'wantJ' and 'wantJ' in opt_literal), f
    assert "error('DBAT:bundle:badInput','Unknown field s.bundle.hip.%s'" in m
    # the gateway hands them on: no literal device / rank any more
    assert not re.search(r'pb\.device\s*=\s*0', cpp) and not re.search(r'pb\.shard_count\s*=\s*1', cpp)
    for call in ('dbat_hip_comm_init', 'dbat_hip_comm_unique_id', 'dbat_hip_set_deterministic', 'dbat_hip_jacobian_csc',
                 'mxCreateSparse', 'opt.term_fun = call_term', 'opt.veto_fun = call_veto', 'mexCallMATLABWithTrap'):
        assert call in cpp, call
    # a failed run never reports "rank ok": the only unconditional rank = lenX sits in the final else of the post-mortem
    wk = m[m.index('E.weakness=struct('):]
    assert wk.count('E.weakness.numerical.rank=lenX') == 1
    assert re.search(r"elseif code==-2 \|\| code==-4\s*\n[^\n]*\n\s*E\.weakness\.numerical\.rank=nan", wk)


def test_plan_reuse_pmdof_and_live_trace_in_the_gateway():
    """Round 6: the gateway keeps its one-rank handle between calls (mexLock / mexAtExit; structure key + dbat_hip_set_values:
    bundle.m:156-159 keeps the serial indices), never raises with a live handle, prints the LSA function's 'trace' line
    from inside the loop, and bundle_hip.m puts the covariance blocks on E.s0's scale when 'pmdof' changes the degrees of
    freedom (the gateway scales with its own sigma0)."""
    cpp = open(GATEWAY).read()
    m = open(WRAPPER).read()
    for call in ('dbat_hip_structure_key', 'dbat_hip_handle_key', 'dbat_hip_set_values', 'mexLock()', 'mexAtExit(drop_kept)',
                 'opt.trace_fun = print_trace', "\"clear\""):
        assert call in cpp, call
    # every mexErrMsgIdAndTxt after the handle exists is preceded by its destruction (or the handle is not alive yet)
    body = cpp[cpp.index('dbat_hip_handle *h = acquire(pb);'):]
    for mt in re.finditer(r'mexErrMsgIdAndTxt\(', body):
        before = body[:mt.start()].rstrip()
        ok = before.endswith('if (!h)') or re.search(r'dbat_hip_destroy\(h\);\s*$', before)
        assert ok, body[max(0, mt.start() - 160):mt.start() + 60]
    assert 'return nullptr' in cpp[cpp.index('static mxArray *sparse_jacobian'):cpp.index('struct MatlabFn')]
    assert "sc=(s0/s0gw)^2" in m and 's0gw=s0;' in m and "'liveTrace',doTrace" in m
    tail = m[m.index('=dbat_hip_mex(P,opt);'):]
    assert tail.index('s0=sqrt(') < tail.index('sc=(s0/s0gw)^2')      # the rescaling uses the pmdof sigma0
