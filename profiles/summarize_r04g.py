#!/usr/bin/env python3
"""The canonical chart as it ships (profiles/collect_r04g.sh: after the fix of the third form's minimum-norm part): soaks, kernel
statistics, percentiles, traffic and counters replace the earlier round-4 files of that chart."""
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import summarize_r04 as s4                                   # noqa: E402

SRC = os.path.join(os.path.dirname(HERE), 'gpurun_out', 'prof_r04g')
s4.SRC = SRC


def main():
    shutil.copy(s4.one('stats_canonical/**/*kernel_stats.csv'), os.path.join(HERE, 'r04_rocprofv3_kernel_stats_canonical.csv'))
    shutil.copy(s4.one('stats_planar_canonical/**/*kernel_stats.csv'), os.path.join(HERE, 'r04_rocprofv3_kernel_stats_planar_canonical.csv'))
    lines = ['# Round 4, the build that ships: the canonical chart\'s step kernels (third form with z by forward substitution)', '',
             '| workload | kernel | FETCH KB | WRITE KB | bytes / launch | algorithmic | ratio | VALU instr / wave | SALU / launch | quad-cycles / wave | waiting | us under the counters |',
             '|---|---|---|---|---|---|---|---|---|---|---|---|']
    for w, algo, fn in (('0_8192_iiwa_canonical_kinematic', 400 * 8192, 'traffic_iiwa_canonical.json'),
                        ('0_8192_planar_canonical_kinematic', 220 * 8192, 'traffic_planar_canonical.json')):
        f = s4.agg(s4.one('pmc_fetch_%s/**/*counter_collection.csv' % w))
        wr = s4.agg(s4.one('pmc_write_%s/**/*counter_collection.csv' % w))
        sq = s4.agg(s4.one('pmc_sq_%s/**/*counter_collection.csv' % w))
        tot = (f['FETCH_SIZE'] + wr['WRITE_SIZE']) * 1024
        json.dump({'kernel': f['_kernel'], 'workload': w, 'FETCH_SIZE_KB': f['FETCH_SIZE'], 'WRITE_SIZE_KB': wr['WRITE_SIZE'],
                   'hbm_bytes_per_launch': tot, 'algorithmic_bytes_per_launch': algo,
                   'note': 'round 4, the build that ships; rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes '
                           '(profiles/collect_r04g.sh, profiles/tools/gpu_pmc_target.py), mean of 20 launches; raw counter x 1024'},
                  open(os.path.join(HERE, fn), 'w'), indent=1)
        lines.append('| %s | `%s` | %.1f | %.1f | %.0f | %d | %.2f | %.0f | %.0f | %.0f | %.2f | %.1f |' % (
            w, f['_kernel'].replace('void atacom::', '')[:48], f['FETCH_SIZE'], wr['WRITE_SIZE'], tot, algo, tot / algo,
            sq['SQ_INSTS_VALU'] / sq['SQ_WAVES'], sq['SQ_INSTS_SALU'], sq['SQ_WAVE_CYCLES'] / sq['SQ_WAVES'],
            sq['SQ_WAIT_ANY'] / sq['SQ_WAVE_CYCLES'], sq['_dur_us']))
    open(os.path.join(HERE, 'r04_pmc_summary_canonical.md'), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
    for n, dst in (('sens_soak_canonical_l4', None), ('sens_soak_canonical_l8', None), ('soak_canonical_f64_l8', None),
                   ('lanes_vs_batch_canonical', None), ('launch_percentiles_canonical', None), ('gpu_suite', None), ('smoke', None)):
        p = os.path.join(SRC, n + '.log')
        if os.path.exists(p):
            shutil.copy(p, os.path.join(HERE, 'r04_' + (dst or n) + '.log'))
    for n in ('bench_default', 'bench_driver_cmd'):
        shutil.copy(os.path.join(SRC, n + '.json'), os.path.join(HERE, 'r04_' + n + '.json'))


if __name__ == '__main__':
    main()
