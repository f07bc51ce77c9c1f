"""Copy the rocprofv3 summaries that gpurun merged into gpurun_out/ to profiles/<round>/<tag>_* and rebuild
profiles/<round>/hbm_traffic.json (the per-launch HBM bytes bench.py reports as roofline.traffic).
usage: python tools/collect_profiles.py r01 f_merged_layer"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if 'nnr::' in r['Kernel_Name']:
            a[r['Kernel_Name'].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in a.items()}


if __name__ == '__main__':
    rnd, tag = sys.argv[1], sys.argv[2]
    src, dst = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles', rnd)
    os.makedirs(dst, exist_ok=True)
    for rel, name in (('prof/bench/bench_kernel_stats.csv', 'bench_kernel_stats.csv'), ('prof/headline/headline_kernel_stats.csv', 'bench_headline_only_kernel_stats.csv'),
                      ('bench_headline_only.txt', 'bench_headline_only.json.txt'), ('prof/pmc1/pmc1_counter_collection.csv', 'pmc_sq.csv'),
                      ('prof/pmc3/pmc3_counter_collection.csv', 'pmc_fetch.csv'), ('prof/pmc4/pmc4_counter_collection.csv', 'pmc_write.csv'),
                      ('timelines.txt', 'timeline.txt'), ('bench.txt', 'bench.json.txt'), ('pmcv/product.txt', 'pmc_clock.txt')):
        p = os.path.join(src, rel)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, f'{tag}_{name}'))
    if len(sys.argv) > 3:     # python tools/collect_profiles.py r02 z_final bf16_4096x128: the second shape of the round
        name = sys.argv[3]
        R, N = (int(x) for x in name.split('_')[1].split('x'))
        for rel, out_name in ((f'prof/{name}_fetch/{name}_fetch_counter_collection.csv', f'{tag}_{name}_pmc_fetch.csv'),
                              (f'prof/{name}_write/{name}_write_counter_collection.csv', f'{tag}_{name}_pmc_write.csv'),
                              (f'prof/{name}_stats/{name}_stats_kernel_stats.csv', f'{tag}_{name}_kernel_stats.csv')):
            shutil.copy(os.path.join(src, rel), os.path.join(dst, out_name))
        f, w = agg(os.path.join(dst, f'{tag}_{name}_pmc_fetch.csv')), agg(os.path.join(dst, f'{tag}_{name}_pmc_write.csv'))
        out = {k: {'fetch_bytes': 2 * f[k]['FETCH_SIZE'] * 1024, 'write_bytes': w[k]['WRITE_SIZE'] * 1024} for k in f if k in w}
        json.dump({'source': f'profiles/{rnd}/{tag}_{name}_pmc_fetch.csv + _pmc_write.csv: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc '
                             f'WRITE_SIZE (separate passes) -- python tools/profile_kernels.py 2 {R} {N} ' + ('bf16' if name.startswith('bf16') else '') +
                             '; FETCH_SIZE doubled (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section); bytes per launch',
                   'shape': [R, N], 'bf16': name.startswith('bf16'), 'kernels': out},
                  open(os.path.join(dst, f'hbm_traffic_{name}.json'), 'w'), indent=1)
        for k, v in out.items():
            if 'mlp' in k or 'wgrad' in k:
                print('%-60s fetch %.3f GB write %.3f GB' % (k[:60], v['fetch_bytes'] / 1e9, v['write_bytes'] / 1e9))
        sys.exit(0)
    f, w = agg(os.path.join(dst, f'{tag}_pmc_fetch.csv')), agg(os.path.join(dst, f'{tag}_pmc_write.csv'))
    out = {k: {'fetch_bytes': 2 * f[k]['FETCH_SIZE'] * 1024, 'write_bytes': w[k]['WRITE_SIZE'] * 1024} for k in f if k in w}
    json.dump({'source': f'profiles/{rnd}/{tag}_pmc_fetch.csv + {tag}_pmc_write.csv: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE '
                         '(separate passes) -- python tools/profile_kernels.py 2; FETCH_SIZE doubled (gfx950: 128-B requests tallied at 64 B, '
                         'MI355X_MICROARCH.md HBM section); bytes per launch, 1024 rays x 192 samples, D=256',
               'shape': [1024, 192], 'bf16': False, 'kernels': out}, open(os.path.join(dst, 'hbm_traffic.json'), 'w'), indent=1)
    for k, v in out.items():
        if 'mlp' in k or 'wgrad' in k:
            print('%-60s fetch %.3f GB write %.3f GB' % (k[:60], v['fetch_bytes'] / 1e9, v['write_bytes'] / 1e9))
