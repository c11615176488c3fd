#!/usr/bin/env python3
"""Turn the raw output of profiles/collect_r06.sh (gpurun_out/prof_<tag>/) into the tracked round-6 files under profiles/:
kernel statistics of the bench command (configs 4, 3, 2, the float64 headline, the saturation batch), launch percentiles, HBM
traffic per launch (raw counters AND the gfx950-corrected figure, 2 x FETCH_SIZE + WRITE_SIZE), SQ counters, mapping-vs-batch
tables, bench lines.
    python profiles/summarize_r06.py [tag]"""
import csv
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import summarize_r04 as s4                                   # noqa: E402  (one(), agg())

TAG = sys.argv[1] if len(sys.argv) > 1 else 'r06'
SRC = os.path.join(os.path.dirname(HERE), 'gpurun_out', 'prof_' + TAG)
s4.SRC = SRC
# algorithmic bytes per launch (SURVEY 8d per-unit figure x environments x element size)
ALGO = {'iiwa': 400 * 8192, 'planar': 220 * 8192, 'circle': 60 * 4096, 'iiwa_f64': 800 * 8192, 'iiwa_65536': 400 * 65536}
W = {'iiwa': '0_8192_iiwa_reference_kinematic_f32', 'planar': '0_8192_planar_reference_kinematic_f32',
     'circle': '0_4096_circle_reference_kinematic_f32', 'iiwa_f64': '0_8192_iiwa_reference_kinematic_f64',
     'iiwa_65536': '0_65536_iiwa_reference_kinematic_f32'}
COLS = ('iiwa', 'planar', 'circle', 'iiwa_f64', 'iiwa_65536')


def main():
    out = ['# Round 6: rocprofv3 summaries of the step kernels (profiles/collect_r06.sh, tag %s)' % TAG, '',
           '## Kernel trace (`rocprofv3 --kernel-trace --stats`)', '',
           '| file | command | kernel | calls | average us |', '|---|---|---|---|---|']
    for tag, dst, cmd, pats in (
            ('stats', 'r06_rocprofv3_kernel_stats.csv', '`bench.py --steps 300 --warmup 30 --min-time 0.3`', ('k_step', 'k_rollout')),
            ('stats_planar', 'r06_rocprofv3_kernel_stats_planar.csv', '`bench.py --env planar ...`', ('k_step',)),
            ('stats_circle', 'r06_rocprofv3_kernel_stats_circle.csv', '`bench.py --env circle --batch 4096 ...`', ('k_step',)),
            ('stats_f64', 'r06_rocprofv3_kernel_stats_f64.csv', '`MB_DTYPE=f64 gpu_microbench.py iiwa` (8192 environments)', ('k_step', 'k_rollout')),
            ('stats_sat', 'r06_rocprofv3_kernel_stats_65536.csv', '`gpu_microbench.py iiwa` at 65536 environments', ('k_step', 'k_rollout'))):
        shutil.copy(s4.one(tag + '/**/*kernel_stats.csv'), os.path.join(HERE, dst))
        rows = list(csv.DictReader(open(os.path.join(HERE, dst))))
        for pat in pats:
            for r in [r for r in rows if pat + '<' in r['Name'] or (pat == 'k_rollout' and 'k_rollout' in r['Name'])][:2 if pat == 'k_rollout' else 1]:
                out.append('| %s | %s | `%s` | %s | %.3f |' % (dst, cmd, r['Name'].split('(')[0][:78], r['Calls'], float(r['AverageNs']) / 1e3))
    out += ['', '## HBM traffic and SQ counters per launch (separate `--pmc` passes with `--kernel-trace` only, mean of 20 launches)', '',
            '| | iiwa 8192 f32 | planar 8192 | circle 4096 | iiwa 8192 FLOAT64 | iiwa 65536 f32 (one lane per env) |', '|---|---|---|---|---|---|']
    rows = {}
    for name, w in W.items():
        f = s4.agg(s4.one('pmc_fetch_%s/**/*counter_collection.csv' % w))
        wr = s4.agg(s4.one('pmc_write_%s/**/*counter_collection.csv' % w))
        sq = s4.agg(s4.one('pmc_sq_%s/**/*counter_collection.csv' % w))
        raw = (f['FETCH_SIZE'] + wr['WRITE_SIZE']) * 1024
        cor = (2 * f['FETCH_SIZE'] + wr['WRITE_SIZE']) * 1024
        json.dump({'kernel': f['_kernel'], 'workload': w, 'FETCH_SIZE_KB': f['FETCH_SIZE'], 'WRITE_SIZE_KB': wr['WRITE_SIZE'],
                   'hbm_bytes_per_launch': raw, 'hbm_bytes_per_launch_corrected': cor, 'algorithmic_bytes_per_launch': ALGO[name],
                   'note': 'round 6 (tag %s): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes '
                           '(profiles/collect_r06.sh, profiles/tools/gpu_pmc_target.py), mean of 20 launches; hbm_bytes_per_launch = raw '
                           'counters x 1024; _corrected = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction of '
                           'MI355X_MICROARCH.md (HBM section) for 16 B / lane streaming reads -- the figure bench.py reports' % TAG},
                  open(os.path.join(HERE, 'traffic_%s.json' % name), 'w'), indent=1)
        rows[name] = (f, wr, sq, raw, cor)

    def row(label, fn):
        out.append('| %s | ' % label + ' | '.join(fn(*rows[n], n) for n in COLS) + ' |')
    row('kernel', lambda f, wr, sq, raw, cor, n: '`%s`' % f['_kernel'][:52])
    row('FETCH_SIZE / WRITE_SIZE (KB)', lambda f, wr, sq, raw, cor, n: '%.1f / %.1f' % (f['FETCH_SIZE'], wr['WRITE_SIZE']))
    row('bytes per launch: raw counters', lambda f, wr, sq, raw, cor, n: '%.0f' % raw)
    row('bytes per launch: 2 x FETCH + WRITE (gfx950)', lambda f, wr, sq, raw, cor, n: '%.0f' % cor)
    row('algorithmic bytes per launch', lambda f, wr, sq, raw, cor, n: '%d' % ALGO[n])
    row('corrected traffic / algorithmic', lambda f, wr, sq, raw, cor, n: '%.2f' % (cor / ALGO[n]))
    row('SQ_WAVES', lambda f, wr, sq, raw, cor, n: '%.0f' % sq['SQ_WAVES'])
    row('SQ_INSTS_VALU per wave', lambda f, wr, sq, raw, cor, n: '%.0f' % (sq['SQ_INSTS_VALU'] / sq['SQ_WAVES']))
    row('SQ_INSTS_SALU per wave', lambda f, wr, sq, raw, cor, n: '%.0f' % (sq['SQ_INSTS_SALU'] / sq['SQ_WAVES']))
    row('wave cycles (x 4 clk) per wave', lambda f, wr, sq, raw, cor, n: '%.0f' % (sq['SQ_WAVE_CYCLES'] / sq['SQ_WAVES']))
    row('clocks per VALU instruction', lambda f, wr, sq, raw, cor, n: '%.2f' % (4 * sq['SQ_WAVE_CYCLES'] / sq['SQ_INSTS_VALU']))
    row('share of wave cycles waiting', lambda f, wr, sq, raw, cor, n: '%.2f' % (sq['SQ_WAIT_ANY'] / sq['SQ_WAVE_CYCLES']))
    row('kernel duration under the counters (us)', lambda f, wr, sq, raw, cor, n: '%.1f' % sq['_dur_us'])
    out.append('')
    open(os.path.join(HERE, 'r06_pmc_summary.md'), 'w').write('\n'.join(out) + '\n')
    print('\n'.join(out))
    for n, dst in (('launch_percentiles.log', 'r06_launch_percentiles.log'), ('lanes_vs_batch_reference.log', 'r06_lanes_vs_batch_reference.log'),
                   ('lanes_vs_batch_f64.log', 'r06_lanes_vs_batch_f64.log'),
                   ('bench_default.json', 'r06_bench_default.json'), ('bench_driver_cmd.json', 'r06_bench_driver_cmd.json'),
                   ('bench_default.time', 'r06_bench_default.time'), ('bench_driver_cmd.time', 'r06_bench_driver_cmd.time')):
        p = os.path.join(SRC, n)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(HERE, dst))


if __name__ == '__main__':
    main()
