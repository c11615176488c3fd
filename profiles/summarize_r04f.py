#!/usr/bin/env python3
"""The build that ships (profiles/collect_r04f.sh): the headline's kernel is now the 8-lane one -- its statistics, traffic and
counters replace the quad kernel's in the tracked files (the quad's stay as *_quad)."""
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import summarize_r04 as s4                                   # noqa: E402

SRC = os.path.join(os.path.dirname(HERE), 'gpurun_out', 'prof_r04f')
s4.SRC = SRC


def main():
    for old, new in (('r04_rocprofv3_kernel_stats.csv', 'r04_rocprofv3_kernel_stats_quad.csv'), ('traffic_iiwa.json', 'traffic_iiwa_quad.json')):
        if os.path.exists(os.path.join(HERE, old)) and not os.path.exists(os.path.join(HERE, new)):
            shutil.copy(os.path.join(HERE, old), os.path.join(HERE, new))
    shutil.copy(s4.one('stats/**/*kernel_stats.csv'), os.path.join(HERE, 'r04_rocprofv3_kernel_stats.csv'))
    w = '0_8192_iiwa_reference_kinematic'
    f = s4.agg(s4.one('pmc_fetch_%s/**/*counter_collection.csv' % w))
    wr = s4.agg(s4.one('pmc_write_%s/**/*counter_collection.csv' % w))
    sq = s4.agg(s4.one('pmc_sq_%s/**/*counter_collection.csv' % w))
    tot = (f['FETCH_SIZE'] + wr['WRITE_SIZE']) * 1024
    algo = 400 * 8192
    json.dump({'kernel': f['_kernel'], 'workload': w, 'FETCH_SIZE_KB': f['FETCH_SIZE'], 'WRITE_SIZE_KB': wr['WRITE_SIZE'],
               'hbm_bytes_per_launch': tot, 'algorithmic_bytes_per_launch': algo,
               'note': 'round 4, the build that ships (single steps on 8 lanes per environment at this batch); rocprofv3 --pmc FETCH_SIZE '
                       'and --pmc WRITE_SIZE in separate passes (profiles/collect_r04f.sh, profiles/tools/gpu_pmc_target.py), mean of 20 '
                       'launches; raw counter x 1024 (with the gfx950 x2 FETCH_SIZE correction of MI355X_MICROARCH.md for 16 B / lane '
                       'streams the fetch side doubles: %.0f bytes per launch in total).' % ((2 * f['FETCH_SIZE'] + wr['WRITE_SIZE']) * 1024)},
              open(os.path.join(HERE, 'traffic_iiwa.json'), 'w'), indent=1)
    lines = ['# Round 4, the build that ships: the headline step kernel (8 lanes per environment, 8192 environments, reference chart)', '',
             '| | value |', '|---|---|',
             '| kernel | `%s` |' % f['_kernel'][:80],
             '| FETCH_SIZE / WRITE_SIZE per launch | %.1f KB / %.1f KB = %.0f bytes (algorithmic %d: ratio %.2f) |' % (
                 f['FETCH_SIZE'], wr['WRITE_SIZE'], tot, algo, tot / algo),
             '| SQ_WAVES | %.0f |' % sq['SQ_WAVES'],
             '| SQ_INSTS_VALU per wave | %.0f |' % (sq['SQ_INSTS_VALU'] / sq['SQ_WAVES']),
             '| SQ_INSTS_SALU per launch | %.0f |' % sq['SQ_INSTS_SALU'],
             '| wave cycles (x4 clk) per wave | %.0f |' % (sq['SQ_WAVE_CYCLES'] / sq['SQ_WAVES']),
             '| share of wave cycles waiting | %.2f |' % (sq['SQ_WAIT_ANY'] / sq['SQ_WAVE_CYCLES']),
             '| kernel duration under the counters | %.1f us |' % sq['_dur_us'], '']
    open(os.path.join(HERE, 'r04_pmc_summary_headline.md'), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
    for n in ('ab_lanes_bench', 'rigid_body', 'launch_percentiles', 'gpu_suite', 'smoke'):
        p = os.path.join(SRC, n + '.log')
        if os.path.exists(p):
            dst = 'r04_' + n + ('_box2' if n == 'ab_lanes_bench' else '') + ('_headline' if n == 'launch_percentiles' else '') + '.log'
            shutil.copy(p, os.path.join(HERE, dst))
    for n in ('bench_default', 'bench_driver_cmd'):
        shutil.copy(os.path.join(SRC, n + '.json'), os.path.join(HERE, 'r04_' + n + '.json'))


if __name__ == '__main__':
    main()
