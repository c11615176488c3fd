#!/usr/bin/env python3
"""Last pass of round 4 (profiles/collect_r04e.sh, final build): the tracked files that pass replaces or adds."""
import glob
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import summarize_r04 as s4                                   # noqa: E402

SRC = os.path.join(os.path.dirname(HERE), 'gpurun_out', 'prof_r04e')
s4.SRC = SRC


def main():
    shutil.copy(s4.one('stats_dyn/**/*kernel_stats.csv'), os.path.join(HERE, 'r04_rocprofv3_kernel_stats_dyn.csv'))
    w = '0_8192_iiwa_reference_rigid_body_ff'
    f = s4.agg(s4.one('pmc_fetch_%s/**/*counter_collection.csv' % w))
    wr = s4.agg(s4.one('pmc_write_%s/**/*counter_collection.csv' % w))
    sq = s4.agg(s4.one('pmc_sq_%s/**/*counter_collection.csv' % w))
    tot = (f['FETCH_SIZE'] + wr['WRITE_SIZE']) * 1024
    algo = 448 * 8192
    json.dump({'kernel': f['_kernel'], 'workload': w, 'FETCH_SIZE_KB': f['FETCH_SIZE'], 'WRITE_SIZE_KB': wr['WRITE_SIZE'],
               'hbm_bytes_per_launch': tot, 'algorithmic_bytes_per_launch': algo,
               'note': 'round 4, final build (the solver state parked in LDS across the dynamics: no scratch traffic); rocprofv3 --pmc '
                       'FETCH_SIZE / WRITE_SIZE in separate passes (profiles/collect_r04e.sh), mean of 20 launches; raw counter x 1024'},
              open(os.path.join(HERE, 'traffic_iiwa_dyn.json'), 'w'), indent=1)
    lines = ['# Round 4, final build: the rigid-body step kernel (quad mapping, 8192 environments, dynamics_mode 2)', '',
             '| | value |', '|---|---|',
             '| kernel | `%s` |' % f['_kernel'][:80],
             '| FETCH_SIZE / WRITE_SIZE per launch | %.1f KB / %.1f KB = %.0f bytes (algorithmic %d: ratio %.2f) |' % (
                 f['FETCH_SIZE'], wr['WRITE_SIZE'], tot, algo, tot / algo),
             '| SQ_INSTS_VALU per wave | %.0f |' % (sq['SQ_INSTS_VALU'] / sq['SQ_WAVES']),
             '| SQ_INSTS_SALU per launch | %.0f |' % sq['SQ_INSTS_SALU'],
             '| wave cycles (x4 clk) per wave | %.0f |' % (sq['SQ_WAVE_CYCLES'] / sq['SQ_WAVES']),
             '| share of wave cycles waiting | %.2f |' % (sq['SQ_WAIT_ANY'] / sq['SQ_WAVE_CYCLES']),
             '| kernel duration under the counters | %.1f us |' % sq['_dur_us'], '']
    open(os.path.join(HERE, 'r04_pmc_summary_dyn.md'), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
    for n in ('ab_dyn_park', 'rigid_body', 'ab_lanes_bench', 'planar_canonical', 'sens_soak_canonical_l4', 'sens_soak_canonical_l8',
              'soak_canonical_f64_l8', 'gpu_suite', 'smoke'):
        p = os.path.join(SRC, n + '.log')
        if os.path.exists(p):
            shutil.copy(p, os.path.join(HERE, 'r04_' + n + '.log'))
    for n in ('bench_default', 'bench_driver_cmd'):
        shutil.copy(os.path.join(SRC, n + '.json'), os.path.join(HERE, 'r04_' + n + '.json'))


if __name__ == '__main__':
    main()
