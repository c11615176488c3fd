#!/usr/bin/env python3
"""Turn the raw output of profiles/collect_r02.sh (gpurun_out/prof_r02/) into the tracked round-2 files under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.path.dirname(HERE), 'gpurun_out', 'prof_r02')


def one(pattern):
    m = sorted(glob.glob(os.path.join(SRC, pattern), recursive=True))
    assert m, pattern
    return m[0]


def agg(path, kern='k_step'):
    a = collections.defaultdict(list)
    dur = []
    for r in csv.DictReader(open(path)):
        if kern in r['Kernel_Name']:
            a[r['Counter_Name']].append(float(r['Counter_Value']))
            if r.get('Start_Timestamp'):
                dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    out = {k: sum(v) / len(v) for k, v in a.items()}
    out['_dur_us'] = sum(dur) / max(len(dur), 1)
    return out


def main():
    shutil.copy(one('stats/**/*kernel_stats.csv'), os.path.join(HERE, 'r02_rocprofv3_kernel_stats.csv'))
    print(open(os.path.join(HERE, 'r02_rocprofv3_kernel_stats.csv')).read().splitlines()[:3])
    sq = {L: agg(one('pmc_sq_l%d/**/*counter_collection.csv' % L)) for L in (4, 8)}
    gr = {L: agg(one('pmc_grbm_l%d/**/*counter_collection.csv' % L)) for L in (4, 8)}
    names = ['SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_ANY',
             'SQ_WAIT_INST_ANY', 'SQ_BUSY_CYCLES']
    print('| counter (per launch) | one env per DPP quad (4 lanes) | one env per 8 lanes |')
    print('|---|---|---|')
    for n in names:
        print('| %s | %.0f | %.0f |' % (n, sq[4].get(n, float('nan')), sq[8].get(n, float('nan'))))
    print('| GRBM_GUI_ACTIVE | %.0f | %.0f |' % (gr[4]['GRBM_GUI_ACTIVE'], gr[8]['GRBM_GUI_ACTIVE']))
    print('| kernel duration under the counters (us) | %.1f | %.1f |' % (sq[4]['_dur_us'], sq[8]['_dur_us']))
    for L in (4, 8):
        w = sq[L]['SQ_WAVES']
        print('lanes=%d per wave: VALU %.0f SALU %.0f wave quad-cycles %.0f (= %.0f clk) parked %.0f' % (
            L, sq[L]['SQ_INSTS_VALU'] / w, sq[L]['SQ_INSTS_SALU'] / w, sq[L]['SQ_WAVE_CYCLES'] / w,
            4 * sq[L]['SQ_WAVE_CYCLES'] / w, sq[L]['SQ_WAIT_ANY'] / w))
    f = agg(one('pmc_fetch/**/*counter_collection.csv'))['FETCH_SIZE']
    w = agg(one('pmc_write/**/*counter_collection.csv'))['WRITE_SIZE']
    traffic = {
        'kernel': 'atacom::k_step<float, Iiwa, 4, true, false> (B=8192)',
        'FETCH_SIZE_KB': f, 'WRITE_SIZE_KB': w, 'hbm_bytes_per_launch': (f + w) * 1024,
        'note': 'round 2; rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/collect_r02.sh, '
                'profiles/tools/gpu_pmc_target.py), mean of 20 launches; raw counter x 1024. The gfx950 x2 FETCH_SIZE '
                'correction of MI355X_MICROARCH.md applies to 16 B/lane streams; these loads are 4 B/lane so the '
                'raw value is reported. Algorithmic bytes per launch: 400 B x 8192 = 3.28 MB.'}
    json.dump(traffic, open(os.path.join(HERE, 'traffic_iiwa.json'), 'w'), indent=1)
    print(json.dumps(traffic)[:160])
    for n in ('bench_default', 'bench_driver_cmd', 'bench_gloo2'):
        shutil.copy(os.path.join(SRC, n + '.json'), os.path.join(HERE, 'r02_' + n + '.json'))
    for n in ('lanes_vs_batch', 'rigid_body', 'sens_l4', 'sens_l8'):
        shutil.copy(os.path.join(SRC, n + '.log'), os.path.join(HERE, 'r02_' + n + '.log'))


if __name__ == '__main__':
    main()
