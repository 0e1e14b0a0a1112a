#!/usr/bin/env python
"""Turn the rocprofv3 output of bench/prof_config.sh (gpurun_out/<dir>) into the tracked summary of
one configuration under profiles/: kernel-trace table, HBM traffic from the PMC passes (gfx950
correction of MI355X_MICROARCH.md: reads = 2 x FETCH_SIZE, writes = WRITE_SIZE; rocprofv3 reports
both in KiB), matrix-core activity, the SQ issue / wait / LDS counters per kernel, and the bench line.
Also refreshes the configuration's entry in profiles/traffic.json (read by bench.py).

    python bench/summarise_config.py gpurun_out/prof_r03a_C3 r03a C3 profiles/r03a_c3 'title'
"""
import collections
import csv
import json
import os
import shutil
import sys

src, prefix, cfg, out = sys.argv[1:5]
title = sys.argv[5] if len(sys.argv) > 5 else ''
WORK = {'C1': '100 cams / 10 000 pts / 100 000 obs, fixed IO', 'C2': '1000 cams / 100 000 pts / 1 000 000 obs, self-calibrating',
        'C3': '1000 cams / 1 000 000 pts / 10 000 000 obs, fixed IO',
        'C4': '5000 cams / 5 000 000 pts / 50 000 000 obs, 4 independent self-calibrated IO blocks',
        'roma': "DBAT's romabundledemo: 60 images / 26 321 pts / 90 561 obs, fixed IO (irregular visibility: tile kernels)",
        'roma-selfcal': "DBAT's romabundledemo_selfcal: 60 images / 26 321 pts / 90 561 obs, 9 IO unknowns",
        'camcal': "DBAT's camcaldemo: 21 images / 100 pts / 2 074 obs, 9 IO unknowns"}


def short(name):
    return name.split('(')[0].replace('void ', '')


def last_json(path):
    if not os.path.exists(path):
        return None
    lines = [ln for ln in open(path).read().strip().splitlines() if ln.startswith('{')]
    return json.loads(lines[-1]) if lines else None


def counters(tag):
    path = os.path.join(src, tag, '%s_%s_counter_collection.csv' % (prefix, tag))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg


rows = list(csv.DictReader(open(os.path.join(src, 'kt', prefix + '_kernel_stats.csv'))))
shutil.copy(os.path.join(src, 'kt', prefix + '_kernel_stats.csv'), out + '_kernel_stats.csv')
pmc = {t: counters(t) for t in ('fetch', 'write', 'mfma', 'sq1', 'sq2', 'sq3')}
avg = lambda v: sum(v) / len(v) if v else 0.0
line = last_json(os.path.join(src, 'bench_line.json'))
prof_line = last_json(os.path.join(src, 'bench_stdout.json'))
with open(out + '_summary.md', 'w') as f:
    f.write('# %s\n\nWorkload %s = %s, one MI355X.\n' % (title, cfg, WORK.get(cfg, cfg)))
    f.write('Collected by `bench/prof_config.sh %s`, summarised by `bench/summarise_config.py`.\n\n' % cfg)
    if line:
        f.write('## Bench line (`python bench.py --config %s --no-cpu-baseline`)\n\n```\n%s\n```\n\n' % (cfg, json.dumps(line)))
    f.write('## Kernel trace\n\n`rocprofv3 --kernel-trace --stats -- python3 bench.py --config %s --no-cpu-baseline --no-solve`\n\n' % cfg)
    if prof_line:
        f.write('ms per step of the profiled run: %.3f (build %.3f, factor+solve %.3f, back-substitution %.3f, trial residual %.3f)\n\n'
                % (prof_line['ms_per_step'], prof_line['ms_build_schur'], prof_line['ms_factor_solve'], prof_line['ms_backsub'],
                   prof_line['ms_trial_residual']))
    f.write('| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n')
    for r in rows[:18]:
        f.write('| `%s` | %s | %.3f | %.1f | %.2f |\n' % (short(r['Name'])[:80], r['Calls'], int(r['TotalDurationNs']) / 1e6,
                                                         float(r['AverageNs']) / 1e3, float(r['Percentage'])))
    f.write('\n## HBM traffic (PMC, separate passes)\n\n`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in their own runs. KiB units; gfx950: FETCH_SIZE counts half '
            'of a streaming read, so reads = 2 x FETCH_SIZE x 1024, writes = WRITE_SIZE x 1024 (gathers are over-counted by this, see '
            '`r02_fetch_calibration.md`).\n\n| kernel | launches | read MB (corrected) | write MB | traffic per launch MB |\n|---|---|---|---|---|\n')
    tr = {}
    for k in sorted(pmc['fetch']):
        if 'dbat::' not in k:
            continue
        rd = 2 * avg(pmc['fetch'][k]['FETCH_SIZE']) * 1024 / 1e6
        wr = avg(pmc['write'].get(k, {}).get('WRITE_SIZE', [])) * 1024 / 1e6
        tr[k] = rd + wr
        if rd + wr > 0.5:
            f.write('| `%s` | %d | %.1f | %.1f | %.1f |\n' % (k, len(pmc['fetch'][k]['FETCH_SIZE']), rd, wr, rd + wr))
    f.write('\n## Matrix-core activity\n\n`--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` (MFMA busy summed over 1024 SIMDs, GUI_ACTIVE over 8 XCDs: '
            'busy fraction = (MFMA/1024)/(GUI/8))\n\n| kernel | SQ_VALU_MFMA_BUSY_CYCLES | GRBM_GUI_ACTIVE | matrix pipe busy |\n|---|---|---|---|\n')
    for k in sorted(pmc['mfma']):
        c = pmc['mfma'][k]
        mb, ga = avg(c.get('SQ_VALU_MFMA_BUSY_CYCLES', [])), avg(c.get('GRBM_GUI_ACTIVE', []))
        if mb > 0 and 'dbat::' in k:
            f.write('| `%s` | %.3g | %.3g | %.3f |\n' % (k[:70], mb, ga, (mb / 1024) / (ga / 8) if ga else 0))
    if pmc['sq1'] or pmc['sq2']:
        f.write('\n## SQ counters of the large kernels (per launch, summed over the chip)\n\n'
                'Three passes (`sq1`: wave / wait, `sq2`: instruction mix, `sq3`: memory side). SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles per wave; '
                'WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (MI355X_MICROARCH.md, PMC slots).\n\n')
        names = ['SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS',
                 'SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_LDS_BANK_CONFLICT',
                 'SQ_LDS_IDX_ACTIVE', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INSTS_VMEM', 'SQ_ACTIVE_INST_VMEM', 'SQ_VALU_MFMA_COEXEC_CYCLES']
        ks = [k for k in sorted(set(pmc['sq1']) | set(pmc['sq2'])) if 'dbat::' in k and
              any(s in k for s in ('k_build_sig', 'k_chol_df', 'k_backsub_sig', 'k_cam_normal', 'k_residual_cm', 'k_build_tile', 'k_backsub<', 'k_heavy_z', 'k_heavy_syrk'))]
        f.write('| counter | ' + ' | '.join('`%s`' % k.replace('dbat::', '')[:28] for k in ks) + ' |\n|---|' + '---|' * len(ks) + '\n')
        for n in names:
            vals = []
            for k in ks:
                v = None
                for t in ('sq1', 'sq2', 'sq3'):
                    if n in pmc[t].get(k, {}):
                        v = avg(pmc[t][k][n])
                vals.append('%.4g' % v if v is not None else '')
            if any(vals):
                f.write('| %s | ' % n + ' | '.join(vals) + ' |\n')
        f.write('\nDerived for the Schur kernel:\n\n')
        for k in ks:
            if 'k_build_sig' not in k and 'k_build_tile' not in k and 'k_heavy' not in k:
                continue
            g = lambda n: next((avg(pmc[t][k][n]) for t in ('sq1', 'sq2', 'sq3') if n in pmc[t].get(k, {})), None)
            wc, wa, wi, ac = g('SQ_WAVE_CYCLES'), g('SQ_WAIT_ANY'), g('SQ_WAIT_INST_ANY'), g('SQ_ACTIVE_INST_ANY')
            if wc:
                f.write('* `%s`: of the wave cycles %.1f %% parked (s_waitcnt / barrier), %.1f %% issue-stalled (of which LDS %.1f %%), %.1f %% issuing; '
                        % (k.replace('dbat::', ''), 100 * wa / wc, 100 * wi / wc, 100 * (g('SQ_WAIT_INST_LDS') or 0) / wc, 100 * ac / wc))
            iv, im, il, bc, la = g('SQ_INSTS_VALU'), g('SQ_INSTS_MFMA'), g('SQ_INSTS_LDS'), g('SQ_LDS_BANK_CONFLICT'), g('SQ_LDS_IDX_ACTIVE')
            if iv:
                f.write('%.0f VALU instructions per MFMA, %.2f LDS instructions per MFMA, LDS bank-conflict cycles %.1f %% of the LDS-active cycles.\n'
                        % ((iv - (im or 0)) / max(im or 1, 1), (il or 0) / max(im or 1, 1), 100 * (bc or 0) / max(la or 1, 1)))
tj = json.load(open('profiles/traffic.json')) if os.path.exists('profiles/traffic.json') else {}
cand = [k for k in tr if 'k_build_sig' in k or 'k_build_tile' in k] or [k for k in tr if 'k_build<' in k]
# the kernel the bench line names (heavy / giant points: k_heavy_syrk dominates the build of scenes like camcal)
bl = last_json(os.path.join(src, 'bench_line.json')) or last_json(os.path.join(src, 'bench_stdout.json'))
if bl and bl.get('roofline', {}).get('kernel') == 'k_heavy_syrk':
    cand = [k for k in tr if 'k_heavy_syrk' in k] or cand
if cand:
    kname = max(cand, key=lambda k: tr[k])
    tj[cfg] = {'kernel': kname.split('<')[0].replace('dbat::', ''), 'kernel_full': kname, 'traffic_bytes_per_launch': tr[kname] * 1e6,
               'correction': 'reads = 2 x FETCH_SIZE x 1024 (gfx950), writes = WRITE_SIZE x 1024',
               'source': 'profiles/' + os.path.basename(out) + '_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of the same bench command)'}
    json.dump(tj, open('profiles/traffic.json', 'w'), indent=1)
print(open(out + '_summary.md').read())
