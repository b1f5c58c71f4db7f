"""Two batches in flight inside one engine (include/vegamx.h: vmx_set_lanes): the second lane borrows every static tensor
and owns its workspace and stream.  Per batch the arithmetic is one lane's: chi2 must be bit-for-bit what a single lane
gives, whatever is interleaved - and anything that changes what an evaluation computes must retire the clone."""
import numpy as np
import pytest

from conftest import load_problem, synth_joint_problem

pytestmark = pytest.mark.gpu

VARIED = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'sigma_velo_disp_lorentz_QSO', 'drp_QSO', 'bias_hcd', 'beta_hcd',
          'L0_hcd', 'bao_amp', 'bias_eta_SiII(1260)', 'bias_eta_CIV(eff)']


@pytest.mark.parametrize('tag,batch', [('joint_synth', 256), ('joint_metals', 96)])
def test_two_lanes_equal_one_lane_bitwise(tag, batch):
    import torch
    from vega_amd import VegaInterface, synthetic
    prob = synth_joint_problem() if tag == 'joint_synth' else load_problem(tag)
    vega = VegaInterface(None, problem=prob, max_batch=batch)
    eng = vega.engine
    eng.set_constant_nl_hint(True, gaussian=True)
    dev = torch.device('cuda', 0)
    pools = [torch.from_numpy(synthetic.walkers(eng.low.theta0, eng.names, batch, varied=VARIED, seed=100 + i)).to(dev)
             for i in range(6)]
    one = [torch.zeros(batch, dtype=torch.float64, device=dev) for _ in pools]
    for th, out in zip(pools, one):
        eng.eval_device(th.data_ptr(), batch, out.data_ptr())
    eng.sync()
    assert all(bool(torch.isfinite(o).all()) and bool((o < 1e99).all()) for o in one)
    eng.set_lanes(2)
    for rounds in range(3):
        two = [torch.zeros(batch, dtype=torch.float64, device=dev) for _ in pools]
        for th, out in zip(pools, two):
            eng.eval_device(th.data_ptr(), batch, out.data_ptr())
        eng.sync()
        for a, b in zip(one, two):
            assert torch.equal(a, b)
        # the host entry in between (it waits for the second lane and runs on the first)
        host = vega.chi2_batch(pools[1].cpu().numpy())
        np.testing.assert_array_equal(host, one[1].cpu().numpy())
    # the convenience entry (chunks alternate between the lanes)
    big = torch.cat(pools)
    got = vega.chi2_batch_device(big)
    assert torch.equal(got, torch.cat(one))
    # new data retire the clone: both lanes then see them
    item = next(iter(prob.items.values()))
    name = next(iter(prob.items))
    eng.set_data(name, 1.01 * item.masked_data_vec)
    two = [torch.zeros(batch, dtype=torch.float64, device=dev) for _ in pools]
    for th, out in zip(pools, two):
        eng.eval_device(th.data_ptr(), batch, out.data_ptr())
    eng.sync()
    eng.set_lanes(1)
    ref = [torch.zeros(batch, dtype=torch.float64, device=dev) for _ in pools]
    for th, out in zip(pools, ref):
        eng.eval_device(th.data_ptr(), batch, out.data_ptr())
    eng.sync()
    for a, b, c in zip(two, ref, one):
        assert torch.equal(a, b) and not torch.equal(a, c)
    vega.close()
