"""Development aid: one small evaluation per batch size with the runtime's log on (crash hunting)."""
import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / 'tests'))
import numpy as np
from conftest import synth_joint_problem
from vega_amd import VegaInterface
for batch in (1, 3, 8, 40):
    prob = synth_joint_problem()
    v = VegaInterface(None, problem=prob, max_batch=batch)
    theta = np.tile(v.engine.low.theta0[None, :], (batch, 1))
    print('batch', batch, flush=True)
    c, st, m = v.engine.eval(theta, want_model=True)
    print('  full chain', c[:2], st[:2], flush=True)
    c = v.chi2_batch(theta)
    print('  chi2 only', c[:2], flush=True)
    v.close()
