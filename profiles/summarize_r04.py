#!/usr/bin/env python3
"""Turn the raw output of profiles/collect_r04.sh (gpurun_out/prof_r03/) into the tracked round-4 files under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.path.dirname(HERE), 'gpurun_out', 'prof_r04')
ALGO = {'iiwa': 400 * 8192, 'iiwa_dyn': 448 * 8192, 'planar': 220 * 8192, 'circle': 60 * 4096}


def one(pattern):
    m = sorted(glob.glob(os.path.join(SRC, pattern), recursive=True))
    assert m, pattern
    return m[0]


def agg(path, kern='k_step'):
    a = collections.defaultdict(list)
    dur, names = [], collections.Counter()
    for r in csv.DictReader(open(path)):
        if kern in r['Kernel_Name']:
            a[r['Counter_Name']].append(float(r['Counter_Value']))
            names[r['Kernel_Name'].split('(')[0]] += 1
            if r.get('Start_Timestamp'):
                dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    out = {k: sum(v) / len(v) for k, v in a.items()}
    out['_dur_us'] = sum(dur) / max(len(dur), 1)
    out['_kernel'] = names.most_common(1)[0][0] if names else '?'
    return out


def main():
    lines = []
    for tag, dst in (('stats', 'r04_rocprofv3_kernel_stats.csv'), ('stats_canonical', 'r04_rocprofv3_kernel_stats_canonical.csv'),
                     ('stats_planar', 'r04_rocprofv3_kernel_stats_planar.csv'),
                     ('stats_circle', 'r04_rocprofv3_kernel_stats_circle.csv'), ('stats_dyn', 'r04_rocprofv3_kernel_stats_dyn.csv')):
        shutil.copy(one(tag + '/**/*kernel_stats.csv'), os.path.join(HERE, dst))
        rows = list(csv.DictReader(open(os.path.join(HERE, dst))))
        top = [r for r in rows if 'k_step' in r['Name']][:1]
        for r in top:
            lines.append('| %s | `%s` | %s | %.3f |' % (dst, r['Name'].split('(')[0][:70], r['Calls'],
                                                     float(r['AverageNs']) / 1e3))
    out = ['# Round 4: rocprofv3 summaries (profiles/collect_r04.sh)', '',
           '## Kernel trace (`rocprofv3 --kernel-trace --stats`) of `bench.py --steps 300 --warmup 30`', '',
           '| file | step kernel | calls | average us |', '|---|---|---|---|'] + lines + ['']
    # traffic
    out += ['## HBM traffic of the step kernel (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, mean of 20 launches)', '',
            '| workload | kernel | FETCH KB | WRITE KB | bytes / launch | algorithmic bytes | ratio |', '|---|---|---|---|---|---|---|']
    for w, name, fn in (('0_8192_iiwa_reference_kinematic', 'iiwa', 'traffic_iiwa.json'),
                        ('0_8192_iiwa_canonical_kinematic', 'iiwa', 'traffic_iiwa_canonical.json'),
                        ('0_8192_iiwa_reference_rigid_body_ff', 'iiwa_dyn', 'traffic_iiwa_dyn.json')):
        f = agg(one('pmc_fetch_%s/**/*counter_collection.csv' % w))
        wr = agg(one('pmc_write_%s/**/*counter_collection.csv' % w))
        tot = (f['FETCH_SIZE'] + wr['WRITE_SIZE']) * 1024
        traffic = {'kernel': f['_kernel'], 'workload': w, 'FETCH_SIZE_KB': f['FETCH_SIZE'], 'WRITE_SIZE_KB': wr['WRITE_SIZE'],
                   'hbm_bytes_per_launch': tot, 'algorithmic_bytes_per_launch': ALGO[name],
                   'note': 'round 4; rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes '
                           '(profiles/collect_r04.sh, profiles/tools/gpu_pmc_target.py), mean of 20 launches; raw counter x 1024 '
                           '(with the gfx950 x2 FETCH_SIZE correction of MI355X_MICROARCH.md for 16 B / lane streams the '
                           'fetch side doubles: %.0f bytes per launch in total).' % ((2 * f['FETCH_SIZE'] + wr['WRITE_SIZE']) * 1024)}
        json.dump(traffic, open(os.path.join(HERE, fn), 'w'), indent=1)
        out.append('| %s | `%s` | %.1f | %.1f | %.0f | %d | %.2f |' % (w, f['_kernel'][:60], f['FETCH_SIZE'], wr['WRITE_SIZE'], tot,
                                                                   ALGO[name], tot / ALGO[name]))
    out.append('')
    # SQ counters
    names = ['SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_ANY',
             'SQ_WAIT_INST_ANY', 'SQ_BUSY_CYCLES']
    ws = ['0_8192_iiwa_reference_kinematic', '0_8192_iiwa_canonical_kinematic', '4_8192_iiwa_canonical_kinematic',
          '0_8192_iiwa_reference_rigid_body_ff', '0_8192_planar_canonical_kinematic']
    sq = {w: agg(one('pmc_sq_%s/**/*counter_collection.csv' % w)) for w in ws}
    out += ['## SQ counters per launch of the step kernel (lanes_batch_env_chart)', '', '| counter | ' + ' | '.join(ws) + ' |',
            '|---|' + '---|' * len(ws)]
    for n in names:
        out.append('| %s | ' % n + ' | '.join('%.0f' % sq[w].get(n, float('nan')) for w in ws) + ' |')
    out.append('| kernel duration under the counters (us) | ' + ' | '.join('%.1f' % sq[w]['_dur_us'] for w in ws) + ' |')
    out.append('| VALU instructions per wave | ' + ' | '.join('%.0f' % (sq[w]['SQ_INSTS_VALU'] / sq[w]['SQ_WAVES']) for w in ws) + ' |')
    out.append('| wave cycles (x4 clk) per wave | ' + ' | '.join('%.0f' % (sq[w]['SQ_WAVE_CYCLES'] / sq[w]['SQ_WAVES']) for w in ws) + ' |')
    out.append('| share of wave cycles waiting | ' + ' | '.join('%.2f' % (sq[w]['SQ_WAIT_ANY'] / sq[w]['SQ_WAVE_CYCLES']) for w in ws) + ' |')
    out.append('')
    open(os.path.join(HERE, 'r04_pmc_summary.md'), 'w').write('\n'.join(out) + '\n')
    print('\n'.join(out))
    for n in ('bench_default', 'bench_driver_cmd'):
        shutil.copy(os.path.join(SRC, n + '.json'), os.path.join(HERE, 'r04_' + n + '.json'))
    for n in ('lanes_vs_batch_canonical', 'lanes_vs_batch_reference', 'rigid_body', 'phase_probe', 'ab_chart_form',
              'launch_percentiles', 'sens_soak_canonical_l4', 'sens_soak_canonical_l8', 'soak_canonical_f64_l8',
              'sens_soak_reference_l4', 'gpu_suite', 'smoke'):
        if os.path.exists(os.path.join(SRC, n + '.log')):
            shutil.copy(os.path.join(SRC, n + '.log'), os.path.join(HERE, 'r04_' + n + '.log'))


if __name__ == '__main__':
    main()
