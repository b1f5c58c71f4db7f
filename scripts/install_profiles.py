"""usage: python scripts/install_profiles.py rNN [src_dir]
Copy the summaries of scripts/gpu_refresh_profiles.sh (gpurun_out/refresh/) into profiles/ under the round's names;
the raw per-kernel counter means of the joint + metals and B = 1 passes are reduced to bytes per launch on the way
(FETCH_SIZE / WRITE_SIZE are KiB; FETCH x 2 per the gfx950 note of MI355X_MICROARCH.md's HBM section)."""
import json
import shutil
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
TAG = sys.argv[1]
SRC = Path(sys.argv[2]) if len(sys.argv) > 2 else REPO / 'gpurun_out' / 'refresh'
DST = REPO / 'profiles'

COPIES = {'bench_full.json': f'{TAG}_bench_full.json', 'bench_core.json': f'{TAG}_bench_core.json',
          'kernel_stats.csv': f'{TAG}_bench_core_kernel_stats.csv', 'traffic.json': f'{TAG}_bench_core_traffic.json',
          'quad_pmc.json': f'{TAG}_quad_pmc.json', 'jm_kernel_stats.csv': f'{TAG}_joint_metals_kernel_stats.csv',
          'cm_kernel_stats.csv': f'{TAG}_coefmod2_kernel_stats.csv',
          'gemv_kernel_stats.csv': f'{TAG}_distortion_gemv_kernel_stats.csv', 'b1_kernel_stats.csv': f'{TAG}_single_point_chain_kernel_stats.csv'}
for src, dst in COPIES.items():
    text = (SRC / src).read_text()
    if src.startswith('bench_'):
        text = text.strip().splitlines()[-1] + '\n'         # (the JSON line only)
    (DST / dst).write_text(text)


def reduce(raw_name, how, keep=None):
    raw = json.loads((SRC / raw_name).read_text())
    out = {}
    for kernel, rec in raw.items():
        if keep and not any(k in kernel for k in keep):
            continue
        c = rec.get('counters_mean_per_launch', {})
        entry = {'runs': rec.get('runs')}
        if 'FETCH_SIZE' in c:
            entry['fetch_bytes_per_launch'] = c['FETCH_SIZE'] * 1024 * 2
        if 'WRITE_SIZE' in c:
            entry['write_bytes_per_launch'] = c['WRITE_SIZE'] * 1024
        out[kernel] = entry
    return {'_how': how, 'kernels': out}


jm = reduce('jm_traffic_raw.json', 'FETCH_SIZE / WRITE_SIZE passes (KiB; FETCH x 2 per the gfx950 note of MI355X_MICROARCH.md) of '
            '`bench.py --core-only --workload joint_metals --batch 512 --lanes 1 --no-static-metals`')
(DST / f'{TAG}_joint_metals_traffic.json').write_text(json.dumps(jm, indent=1, sort_keys=True) + '\n')
gemv = reduce('gemv_traffic_raw.json', 'FETCH_SIZE pass of scripts/gpu_matvec_only.py (B = 1 product, 2500^2, 8 distinct matrices '
              'round-robin then one reused); KiB x 2', keep=('k_gemv1',))
(DST / f'{TAG}_distortion_gemv_traffic.json').write_text(json.dumps(gemv, indent=1, sort_keys=True) + '\n')
# the figures DESIGN section 5 quotes, derived from the counter means (1024 SIMDs, 32 shader engines on the part)
pmc = json.loads((SRC / 'quad_pmc.json').read_text())
derived = {}
for tag, needle in (('quadratic_form_product', 'k_gemm_nt44<12'), ('pk_tab2', 'k_pk_tab2<64')):
    name = next(k for k in pmc if needle in k)
    c, runs = pmc[name]['counters_mean_per_launch'], pmc[name]['runs']
    cycles = c['SQ_BUSY_CYCLES'] / 32
    us = runs[0]['avg_us']
    d = {'kernel': name, 'cycles_per_launch': cycles, 'avg_us_of_that_pass': us, 'clock_GHz': cycles / us / 1e3,
         'valu_wave_instructions': c['SQ_INSTS_VALU'], 'valu_issue_slot_fraction': c['SQ_INSTS_VALU'] / 1024 * 4 / cycles,
         'lds_bank_conflict_cycles': c['SQ_LDS_BANK_CONFLICT'], 'wave_cycles': c['SQ_WAVE_CYCLES']}
    if c.get('SQ_INSTS_MFMA'):
        d.update(mfma_instructions=c['SQ_INSTS_MFMA'], mfma_busy_cycles_per_simd=c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024,
                 mfma_busy_fraction=c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cycles)
    derived[tag] = d
pmc = {'derived': derived, **pmc}
# the joint + metals workload's SQ passes, reduced the same way for its large kernels
jm_pmc = json.loads((SRC / 'jm_pmc.json').read_text())
jm_derived = {}
for name, rec in jm_pmc.items():
    c = rec.get('counters_mean_per_launch', {})
    if not c.get('SQ_BUSY_CYCLES') or rec['runs'][0]['avg_us'] < 20:
        continue
    cycles = c['SQ_BUSY_CYCLES'] / 32
    d = {'avg_us': rec['runs'][0]['avg_us'], 'calls': rec['runs'][0]['calls'], 'cycles_per_launch': cycles,
         'clock_GHz': cycles / rec['runs'][0]['avg_us'] / 1e3,
         'valu_issue_slot_fraction': c.get('SQ_INSTS_VALU', 0.0) / 1024 * 4 / cycles,
         'wait_inst_any_fraction_of_wave_cycles': c.get('SQ_WAIT_INST_ANY', 0.0) / max(c.get('SQ_WAVE_CYCLES', 1.0), 1.0)}
    if c.get('SQ_INSTS_MFMA'):
        d['mfma_busy_fraction'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cycles
    jm_derived[name] = d
(DST / f'{TAG}_joint_metals_pmc.json').write_text(json.dumps({'_how': 'SQ counter passes of `bench.py --core-only --workload joint_metals --batch 512 '
                                                              '--lanes 1 --no-static-metals` (1024 SIMDs, 32 shader engines): kernels of 20 us and more',
                                                              'derived': jm_derived, 'raw': jm_pmc}, indent=1) + '\n')
(DST / f'{TAG}_quad_pmc.json').write_text(json.dumps(pmc, indent=1) + '\n')
print('installed', len(COPIES) + 3, 'files into', DST)
