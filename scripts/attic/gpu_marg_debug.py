"""Diagnosis: marginalisation coefficients of the engine vs NumPy on the engine's own model vs the reference fixture."""
import sys, tempfile, pathlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from conftest import marginalization_problem, MARGINALIZATION_CASES, GOLDEN
from vega_amd import VegaInterface
exp = np.load(GOLDEN / 'expected_marg_coeff.npz')
prob = marginalization_problem(pathlib.Path(tempfile.mkdtemp()), MARGINALIZATION_CASES['rtmax'])
vega = VegaInterface(None, problem=prob, max_batch=16)
it = prob.items['lyalya_lyalya']
chi2, coeff = vega.chi2(return_marg_coeff=True)
c = coeff['lyalya_lyalya']
model = vega.compute_model()['lyalya_lyalya']
ref_model = exp['cov/fid/model_plain']
print('model err / scale', np.abs(model - ref_model).max() / np.abs(ref_model).max(), 'abs', np.abs(model-ref_model).max())
diff = it.masked_data_vec - model[it.model_mask]
c_np = it.marg_diff2coeff.dot(diff)
ref = exp['cov/fid/coeff']
print('engine vs numpy-on-engine-model', np.abs(c - c_np).max(), 'engine vs ref', np.abs(c - ref).max(), 'numpy vs ref', np.abs(c_np - ref).max())
bad = np.argsort(-np.abs(c - ref))[:6]
print(bad, (c - ref)[bad], (c_np - ref)[bad])
dref = it.masked_data_vec - ref_model[it.model_mask]
print('numpy on ref model vs ref', np.abs(it.marg_diff2coeff.dot(dref) - ref).max())
