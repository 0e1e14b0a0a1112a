#!/usr/bin/env python
"""Turn the rocprofv3 output of bench/prof_round.sh (gpurun_out/<dir>) into the
tracked summary under profiles/: kernel-trace table, HBM traffic from the PMC
passes (gfx950 correction: reads = 2 x FETCH_SIZE x 1024, calibrated on
k_axpby; writes = WRITE_SIZE x 1024), matrix-core activity, bench lines.

    python bench/summarise_profiles.py gpurun_out/prof_r02 r02 profiles/r02_c3_v1 'title'
"""
import collections, csv, json, os, shutil, sys

src, prefix, out = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(os.path.join(src, 'kt', prefix + '_kernel_stats.csv'))))
shutil.copy(os.path.join(src, 'kt', prefix + '_kernel_stats.csv'), out + '_kernel_stats.csv')
rd = lambda n: open(os.path.join(src, n)).read().strip().splitlines()[-1]
pmc = {}
for tag in ('fetch', 'write', 'mfma'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(src, tag, '%s_%s_counter_collection.csv' % (prefix, tag)))):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    pmc[tag] = agg
title = sys.argv[4] if len(sys.argv) > 4 else ''
with open(out + '_summary.md', 'w') as f:
    f.write('# %s\n\nWorkload C3 = 1000 cams / 1 000 000 pts / 10 000 000 obs, fixed IO, one MI355X.\n' % title)
    f.write('Collected by `bench/prof_round.sh`, summarised by `bench/summarise_profiles.py`.\n\n')
    f.write('## Default bench line (`python bench.py`: 20 steps, 3 warm-up, with cpu_baseline)\n\n```\n%s\n```\n\n' % rd('bench_default.json'))
    if os.path.exists(os.path.join(src, 'bench_c2.json')):
        f.write('C2 (1000 cams / 100k pts / 1M obs, self-calibrating), `python bench.py --config C2 --no-cpu-baseline`:\n\n```\n%s\n```\n\n' % rd('bench_c2.json'))
    f.write('## Kernel trace\n\n`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline`\n'
            '(7 LM steps incl. warm-up + 1 set-up linearisation; raw csv next to this file)\n\nbench.py line of the profiled run:\n\n```\n%s\n```\n\n' % rd('bench_stdout.json'))
    f.write('| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n')
    for r in rows[:20]:
        short = r['Name'].split('(')[0].replace('void ', '')
        if short.startswith('Cijk_'): short = 'rocBLAS dgemm (Tensile) ' + short[:44]
        f.write('| `%s` | %s | %.3f | %.1f | %.2f |\n' % (short[:80], r['Calls'], int(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
    f.write('\n## HBM traffic (PMC, separate passes)\n\n`rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`, the same with `--pmc WRITE_SIZE`.\n'
            'Units: counter x 1024 B. gfx950 correction: FETCH_SIZE counts half of the bytes of a streaming read (calibrated on `k_axpby`, which reads 2 x 24.05 MB and writes 24.05 MB), '
            'so reads = 2 x FETCH_SIZE x 1024, writes = WRITE_SIZE x 1024.\n\n')
    f.write('| kernel | launches | read MB (corrected) | write MB | traffic per launch MB |\n|---|---|---|---|---|\n')
    tr = {}
    for k in sorted(pmc['fetch']):
        if 'dbat::' not in k: continue
        fv = pmc['fetch'][k]['FETCH_SIZE']; wv = pmc['write'].get(k, {}).get('WRITE_SIZE', [0.0])
        rdm = 2 * sum(fv) / len(fv) * 1024 / 1e6; wr = sum(wv) / len(wv) * 1024 / 1e6
        tr[k] = rdm + wr
        if rdm + wr > 1: f.write('| `%s` | %d | %.1f | %.1f | %.1f |\n' % (k, len(fv), rdm, wr, rdm + wr))
    kname = [k for k in tr if 'k_build_sig' in k or 'k_build_tile' in k][0]
    f.write('\nAlgorithmic bytes of `%s` per launch (DESIGN.md 4): 40*no + 24*np + 48*nc + 8*NS^2 = 712 MB (552 MB without the 16 B/observation of the residual that is no longer written); measured %.0f MB.\n' % (kname, tr[kname]))
    f.write('\n## Matrix-core activity (PMC)\n\n`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`\n'
            '(SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: busy fraction of the matrix pipe = (MFMA/1024)/(GUI/8))\n\n'
            '| kernel | launches | SQ_VALU_MFMA_BUSY_CYCLES (avg) | GRBM_GUI_ACTIVE (avg) | matrix pipe busy |\n|---|---|---|---|---|\n')
    for k in sorted(pmc['mfma']):
        c = pmc['mfma'][k]
        if 'SQ_VALU_MFMA_BUSY_CYCLES' not in c: continue
        mb = sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / len(c['SQ_VALU_MFMA_BUSY_CYCLES'])
        ga = sum(c.get('GRBM_GUI_ACTIVE', [0])) / max(1, len(c.get('GRBM_GUI_ACTIVE', [0])))
        if mb > 0 and 'dbat::' in k: f.write('| `%s` | %d | %.3g | %.3g | %.3f |\n' % (k[:70], len(c['SQ_VALU_MFMA_BUSY_CYCLES']), mb, ga, (mb / 1024) / (ga / 8) if ga else 0))
tj = {}
if os.path.exists('profiles/traffic.json'):
    tj = json.load(open('profiles/traffic.json'))
tj['C3'] = {'kernel': kname.split('<')[0].replace('dbat::', ''), 'kernel_full': kname, 'traffic_bytes_per_launch': tr[kname] * 1e6,
            'correction': 'reads = 2 x FETCH_SIZE x 1024 (gfx950), writes = WRITE_SIZE x 1024',
            'source': 'profiles/' + os.path.basename(out) + '_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of the same bench command)'}
json.dump(tj, open('profiles/traffic.json', 'w'), indent=1)
for extra in ('bench_c1.json', 'bench_c4.json', 'bench_comm1.json'):
    if os.path.exists(os.path.join(src, extra)):
        with open(out + '_summary.md', 'a') as f:
            f.write('\n`%s`:\n\n```\n%s\n```\n' % (extra, rd(extra)))
