#!/usr/bin/env python3
"""Turn the raw output of profiles/collect_r05d.sh (gpurun_out/prof_<tag>/) into the tracked round-5 files under profiles/:
kernel statistics of the bench command (configs 4, 3, 2), launch percentiles, HBM traffic per launch (raw counters AND the
gfx950-corrected figure, 2 x FETCH_SIZE + WRITE_SIZE), SQ counters, mapping-vs-batch table, bench lines.
    python profiles/summarize_r05.py [tag]"""
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import summarize_r04 as s4                                   # noqa: E402  (one(), agg())

TAG = sys.argv[1] if len(sys.argv) > 1 else 'r05d'
SRC = os.path.join(os.path.dirname(HERE), 'gpurun_out', 'prof_' + TAG)
s4.SRC = SRC
ALGO = {'iiwa': 400 * 8192, 'planar': 220 * 8192, 'circle': 60 * 4096, 'iiwa_dyn': 448 * 8192}
W = {'iiwa': '0_8192_iiwa_reference_kinematic', 'planar': '0_8192_planar_reference_kinematic',
     'circle': '0_4096_circle_reference_kinematic', 'iiwa_dyn': '0_8192_iiwa_reference_rigid_body_ff'}


def main():
    out = ['# Round 5: rocprofv3 summaries of the step kernels (profiles/collect_r05d.sh, tag %s)' % TAG, '',
           '## Kernel trace (`rocprofv3 --kernel-trace --stats`) of `bench.py --steps 300 --warmup 30 --min-time 0.3`', '',
           '| file | step kernel | calls | average us |', '|---|---|---|---|']
    import csv
    for tag, dst in (('stats', 'r05_rocprofv3_kernel_stats.csv'), ('stats_planar', 'r05_rocprofv3_kernel_stats_planar.csv'),
                     ('stats_circle', 'r05_rocprofv3_kernel_stats_circle.csv'), ('stats_dyn', 'r05_rocprofv3_kernel_stats_dyn.csv')):
        shutil.copy(s4.one(tag + '/**/*kernel_stats.csv'), os.path.join(HERE, dst))
        for r in [r for r in csv.DictReader(open(os.path.join(HERE, dst))) if 'k_step' in r['Name']][:1]:
            out.append('| %s | `%s` | %s | %.3f |' % (dst, r['Name'].split('(')[0][:70], r['Calls'], float(r['AverageNs']) / 1e3))
    out += ['', '## HBM traffic and SQ counters per launch (separate `--pmc` passes with `--kernel-trace` only, mean of 20 launches)', '',
            '| | iiwa 8192 | planar 8192 | circle 4096 | iiwa 8192 rigid body (ff) |', '|---|---|---|---|---|']
    rows = {}
    for name, w in W.items():
        f = s4.agg(s4.one('pmc_fetch_%s/**/*counter_collection.csv' % w))
        wr = s4.agg(s4.one('pmc_write_%s/**/*counter_collection.csv' % w))
        sq = s4.agg(s4.one('pmc_sq_%s/**/*counter_collection.csv' % w))
        raw = (f['FETCH_SIZE'] + wr['WRITE_SIZE']) * 1024
        cor = (2 * f['FETCH_SIZE'] + wr['WRITE_SIZE']) * 1024
        json.dump({'kernel': f['_kernel'], 'workload': w, 'FETCH_SIZE_KB': f['FETCH_SIZE'], 'WRITE_SIZE_KB': wr['WRITE_SIZE'],
                   'hbm_bytes_per_launch': raw, 'hbm_bytes_per_launch_corrected': cor, 'algorithmic_bytes_per_launch': ALGO[name],
                   'note': 'round 5 (tag %s): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes '
                           '(profiles/collect_r05d.sh, profiles/tools/gpu_pmc_target.py), mean of 20 launches; hbm_bytes_per_launch = raw '
                           'counters x 1024; _corrected = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction of '
                           'MI355X_MICROARCH.md (HBM section) for 16 B / lane streaming reads -- the figure bench.py reports' % TAG},
                  open(os.path.join(HERE, 'traffic_%s.json' % name), 'w'), indent=1)
        rows[name] = (f, wr, sq, raw, cor)
    def row(label, fn):
        out.append('| %s | ' % label + ' | '.join(fn(*rows[n], n) for n in ('iiwa', 'planar', 'circle', 'iiwa_dyn')) + ' |')
    row('kernel', lambda f, wr, sq, raw, cor, n: '`%s`' % f['_kernel'][:48])
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
    open(os.path.join(HERE, 'r05_pmc_summary.md'), 'w').write('\n'.join(out) + '\n')
    print('\n'.join(out))
    for n, dst in (('launch_percentiles.log', 'r05_launch_percentiles.log'), ('lanes_vs_batch_reference.log', 'r05_lanes_vs_batch_reference.log'),
                   ('bench_default.json', 'r05_bench_default.json'), ('bench_driver_cmd.json', 'r05_bench_driver_cmd.json')):
        p = os.path.join(SRC, n)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(HERE, dst))


if __name__ == '__main__':
    main()
