"""Turn the rocprofv3 passes of `bench.py --core-only` into profiles/rNN_bench_core_traffic.json.
usage: python scripts/gpu_traffic_json.py <out.json> <stats_dir> <fetch_pmc_dir> <write_pmc_dir>

FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for
16-B-per-lane streaming reads on gfx950.  The algorithmic bytes are those of BASELINE configs[2] at B = 256:
the operator once + the batch's input and output vectors."""
import glob, json, sys
import pandas as pd

B = 256
SHAPES_DIST = [(2500, 2500), (5000, 5000)]
SHAPES_COV = [(1590, 1590), (3180, 3180)]
NQ = [2500 + 12, 5000]           # quadratic form: n_model + additive post-distortion broadband coefficients per item
CLASSES = {
    # chi2-only steps: one half-triangle product with Q' per item - half the matrix + the batch's vectors in; the product is
    # contracted in the epilogue (round 2: no vectors out, 16 bytes of partial sums per walker and block are negligible)
    'quadratic_form_product': ('k_gemm_nt44<12,', sum(8 * n * n / 2 + 8 * B * n for n in NQ)),
    'distortion_product': ('k_gemm_nt44<6,', sum(8 * m * n + 8 * B * (m + n) for m, n in SHAPES_DIST)),
    'invcov_product': ('k_gemm_nt<64, 64, 32, 8>', sum(8 * m * n / 2 + 8 * B * (m + n) for m, n in SHAPES_COV)),
    'fftlog_spline_product': ('k_gemm_nt44<2,', None),
    'pk_multipoles': ('k_pk_tab2', None),
    'xi_bins': ('k_xi_quad_plain', None),
    'chi2': ('k_chi2_parts', None),
}


def trace(d):
    f = glob.glob(f'{d}/**/*_kernel_trace.csv', recursive=True)[0]
    kt = pd.read_csv(f)
    kt['dur_us'] = (kt.End_Timestamp - kt.Start_Timestamp) / 1e3
    return kt


def counter(d, name):
    f = glob.glob(f'{d}/**/*_counter_collection.csv', recursive=True)[0]
    df = pd.read_csv(f)
    return df[df.Counter_Name == name]


out_path, d_stats, d_fetch, d_write = sys.argv[1:5]
kt_s, kt_f, kt_w = trace(d_stats), trace(d_fetch), trace(d_write)
fetch, write = counter(d_fetch, 'FETCH_SIZE'), counter(d_write, 'WRITE_SIZE')
kernels = {}
for cls, (needle, algo) in CLASSES.items():
    def pick(df):
        return df[df.Kernel_Name.str.contains(needle, regex=False)]
    s = pick(kt_s)
    if not len(s):
        continue
    fb = float(pick(fetch).Counter_Value.mean()) * 1024 * 2
    wb = float(pick(write).Counter_Value.mean()) * 1024
    kernels[cls] = {'kernel': s.Kernel_Name.iloc[0], 'fetch_bytes_per_launch': fb, 'write_bytes_per_launch': wb,
                    'hbm_bytes_per_launch': fb + wb, 'algorithmic_bytes_per_launch': algo,
                    'avg_us_kernel_trace_run': float(s.dur_us.mean()), 'launches': int(len(s)),
                    'avg_us_pmc_runs': [float(pick(kt_f).dur_us.mean()), float(pick(kt_w).dur_us.mean())]}
json.dump({'_how': __doc__, 'kernels': kernels}, open(out_path, 'w'), indent=1)
print('wrote', out_path, list(kernels))
