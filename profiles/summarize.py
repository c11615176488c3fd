#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of profiles/collect.sh (gpurun_out/prof/) into the tracked files under profiles/:
kernel-stats CSVs, traffic_iiwa.json, and the numbers of r01_pmc_summary.md (printed; pasted into the .md)."""
import collections
import csv
import glob
import json
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.path.dirname(HERE), 'gpurun_out', 'prof')


def one(pattern):
    m = sorted(glob.glob(os.path.join(SRC, pattern), recursive=True))
    assert m, pattern
    return m[0]


def agg(path, kern='k_step'):
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kern in r['Kernel_Name']:
            a[r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in a.items()}


def main():
    for tag, name in (('stats_quad', 'quad'), ('stats_lane', 'lane_per_env')):
        dst = os.path.join(HERE, 'r01_rocprofv3_kernel_stats_%s.csv' % name)
        shutil.copy(one('%s/**/*kernel_stats.csv' % tag), dst)
        print(name, open(dst).read().splitlines()[:3])
    out = {}
    for L in (1, 4):
        out[L] = agg(one('pmc_sq_l%d/**/*counter_collection.csv' % L))
    names = ['SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_ANY',
             'SQ_WAIT_INST_ANY', 'SQ_BUSY_CYCLES']
    print('| counter (per launch) | one env per lane | one env per DPP quad |')
    for n in names:
        print('| %s | %.0f | %.0f |' % (n, out[1].get(n, float('nan')), out[4].get(n, float('nan'))))
    for L in (1, 4):
        w = out[L]['SQ_WAVES']
        print('lanes=%d per wave: VALU %.0f SALU %.0f wave quad-cycles %.0f parked %.0f' % (
            L, out[L]['SQ_INSTS_VALU'] / w, out[L]['SQ_INSTS_SALU'] / w, out[L]['SQ_WAVE_CYCLES'] / w,
            out[L]['SQ_WAIT_ANY'] / w))
    f = agg(one('pmc_fetch/**/*counter_collection.csv'))['FETCH_SIZE']
    w = agg(one('pmc_write/**/*counter_collection.csv'))['WRITE_SIZE']
    traffic = {
        'kernel': 'atacom::k_step<float, Iiwa, 4, true> (B=8192)',
        'FETCH_SIZE_KB': f, 'WRITE_SIZE_KB': w, 'hbm_bytes_per_launch': (f + w) * 1024,
        'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/collect.sh, '
                'profiles/tools/gpu_pmc_target.py), mean of 20 launches; raw counter x 1024. The gfx950 x2 FETCH_SIZE '
                'correction of MI355X_MICROARCH.md applies to 16 B/lane streams; these loads are 4 B/lane so the '
                'raw value is reported. Algorithmic bytes per launch: 400 B x 8192 = 3.28 MB.'}
    json.dump(traffic, open(os.path.join(HERE, 'traffic_iiwa.json'), 'w'), indent=1)
    print(json.dumps(traffic)[:200])
    print(open(os.path.join(SRC, 'bench.json')).read()[:900])
    for t in ('quad', 'lane'):
        print(t, open(os.path.join(SRC, 'bench_under_rocprof_%s.log' % t)).read().strip().splitlines()[-1][:300])


if __name__ == '__main__':
    main()
