"""Correlations whose `[model]` sections disagree on the transform settings: one engine per distinct setting.

In the reference every correlation item owns its operators - `num_bins_muk` sizes its (k, mu) grid
(vega/power_spectrum.py:52-58), `old_fftlog` / `fht_lowring` choose its P(k) -> xi transform (vega/pktoxi.py:36-60), an
`fvoigt_model` names its Voigt-profile table (vega/power_spectrum.py:310-340) - and nothing couples the items before the
chi2 sum (vega/vega_interface.py:232-316).  A vegamx engine holds ONE mu grid, ONE FFTLog operator set and ONE Voigt table
in HBM, which is what every real configuration needs (the settings come from a shared template); here the rare mixed configuration is served by partitioning the items by setting and giving each part its
own engine on the same GPU: chi2 is the sum of the parts' chi2 (the priors enter once, through the first part), a model
is the concatenation of the parts' models in the configured item order.  A global covariance (vega/vega_interface.py:295-304)
couples the parts: each engine takes its diagonal block of the global inverse, the cross terms are formed from the engines'
models on the device (`EngineGroup._eval_global_device`).

``make_engine`` returns a plain ``Engine`` when the settings agree - the group is never on the hot path of BASELINE.json's
configurations.
"""
import copy

import numpy as np

from .engine import Engine

SENTINEL = 1e100


def _fvoigt_key(pipe):
    """The Voigt-profile table of an `fvoigt` HCD model (`fvoigt_model`, vega/power_spectrum.py:310-340; one table per engine),
    as a hashable: None when the pipeline has none."""
    table = getattr(pipe.pk, 'fvoigt_table', None)
    return None if table is None else hash(np.ascontiguousarray(table).tobytes())


def item_settings(item):
    """(num_bins_muk, old_fftlog, fht_lowring, fht_extrap, Fvoigt table) of a correlation item; its metal pipelines read the same
    `[model]` section in the reference (vega/metals.py:60-75), so a disagreement on the transform inside one item is a
    set-up error.  (Pipelines without a Voigt table go with whatever table their item's other pipelines use.)"""
    pipes = [item.core] + [m.pipeline for m in item.metals]
    keys = {(p.pk.n_mu, bool(p.xi.old_fftlog), bool(p.xi.fht_lowring), bool(getattr(p.xi, 'fht_extrap', False)) and not p.xi.old_fftlog)
            for p in pipes}
    if len(keys) != 1:
        raise ValueError(f'the pipelines of one correlation item disagree on num_bins_muk / old_fftlog / fht_lowring / fht_extrap: {sorted(keys)}')
    tables = {k for k in map(_fvoigt_key, pipes) if k is not None}
    if len(tables) > 1:
        raise NotImplementedError('one correlation item with two different Fvoigt tables')
    return keys.pop() + (tables.pop() if tables else None,)


def setting_groups(problem):
    """Item names partitioned by ``item_settings``, groups and members in the configured order."""
    groups = {}
    for name, item in problem.items.items():
        groups.setdefault(item_settings(item), []).append(name)
    # an item without a Voigt table can share an engine with one that has one (same transform settings)
    merged = {}
    for key, names in groups.items():
        if key[-1] is None:
            host = next((k for k in groups if k[:-1] == key[:-1] and k[-1] is not None), key)
            merged.setdefault(host, []).extend(names)
        else:
            merged.setdefault(key, []).extend(names)
    order = {name: i for i, name in enumerate(problem.items)}
    out = [sorted(names, key=order.get) for names in merged.values()]
    return sorted(out, key=lambda names: order[names[0]])


def make_engine(problem, **kwargs):
    groups = setting_groups(problem)
    if len(groups) == 1:
        return Engine(problem, **kwargs)
    return EngineGroup(problem, groups, **kwargs)


class EngineGroup:
    """The ``Engine`` surface over one engine per setting group (see the module text)."""

    def __init__(self, problem, groups, **kwargs):
        self.prob = problem
        self.groups = [list(g) for g in groups]
        self.children = []
        # A global covariance (reference vega/vega_interface.py:295-304: chi2 = r^T G r over the concatenated masked residual)
        # couples the engines: r^T G r = sum_i r_i^T G_ii r_i + 2 sum_{i<j} r_i^T G_ij r_j with the blocks of G over the
        # engines' items.  Engine i takes G_ii as ITS global matrix (priors, blinding and status stay where they are); the
        # cross terms are formed here from the engines' models on the device (`_cross_terms`).
        blocks = None
        if problem.global_cov is not None:
            gm = problem.global_masks()
            rows, off = {}, 0
            for name, item in problem.items.items():
                rows[name] = np.arange(off, off + item.data_size)
                off += item.data_size
            blocks = [np.concatenate([rows[n] for n in names]) for names in self.groups]
            self._global = dict(G=np.asarray(gm['chi2_matrix'], dtype=np.float64), blocks=blocks, dev=None,
                                data={n: np.asarray(it.masked_data_vec, dtype=np.float64).copy() for n, it in problem.items.items()})
        else:
            self._global = None
        try:
            for gi, names in enumerate(self.groups):
                sub = copy.copy(problem)
                sub.items = {n: problem.items[n] for n in names}
                if gi > 0:
                    sub.priors = {}             # (the Gaussian priors are part of the first engine's chi2 only)
                extra = {}
                if blocks is not None:
                    sub.global_cov, sub._global = None, None
                    extra['global_chi2_matrix'] = self._global['G'][np.ix_(blocks[gi], blocks[gi])]
                self.children.append(Engine(sub, **kwargs, **extra))
        except Exception:
            self.close()
            raise
        # (fast_metals / static-basis plans are keyed by item name: every engine picks its own items' entries)
        self.metal_plan = kwargs.get('metal_plan') or {}
        self.static_poly = bool(kwargs.get('static_poly', True))
        first = self.children[0]
        self.lib, self.low, self.names, self.n_params, self.max_batch = first.lib, first.low, first.names, first.n_params, first.max_batch
        assert all(c.names == self.names for c in self.children)
        self.lanes = 1
        self.item_names = list(problem.items)
        self._owner = {n: c for c, names in zip(self.children, self.groups) for n in names}
        # models in the configured item order
        self.model_slices, self._gather = {}, []
        off = 0
        for name in self.item_names:
            c = self._owner[name]
            sl = c.model_slices[name]
            self.model_slices[name] = slice(off, off + sl.stop - sl.start)
            self._gather.append((self.children.index(c), sl, self.model_slices[name]))
            off += sl.stop - sl.start
        self.model_size = off
        self.pipe_index = {key: (ci, pid) for ci, c in enumerate(self.children) for key, pid in c.pipe_index.items()}
        self.metal_source = {key: v for c in self.children for key, v in c.metal_source.items()}
        self.csr_items = [n for c in self.children for n in c.csr_items]
        self._dev_bufs = None

    # ---- evaluation
    def theta_from_params(self, params=None):
        return self.children[0].theta_from_params(params)

    def eval(self, theta, want_model=False):
        theta = np.asarray(theta, dtype=np.float64)
        if theta.ndim == 1:
            theta = theta[None, :]
        if self._global is not None:
            return self._eval_global_host(theta, want_model)
        parts = [c.eval(theta, want_model) for c in self.children]
        status = parts[0][1].copy()
        for p in parts[1:]:
            status |= p[1]
        chi2 = np.where(status != 0, SENTINEL, sum(p[0] for p in parts))
        model = None
        if want_model:
            model = np.empty((theta.shape[0], self.model_size))
            for ci, src, dst in self._gather:
                model[:, dst] = parts[ci][2][:, src]
        return chi2, status, model

    # ---- a global covariance over the engines
    def _global_state(self, device):
        """Device copies of what the cross terms need (made once): per engine the positions of its masked bins in its model
        vector and its masked data, per pair of engines the zero-padded off-diagonal block of G."""
        import torch
        g = self._global
        if g['dev'] is None:
            pad = lambda n: (n + 31) // 32 * 32
            idx, data = [], []
            for c, names in zip(self.children, self.groups):
                pos = [c.model_slices[n].start + np.flatnonzero(self.prob.items[n].model_mask) for n in names]
                idx.append(torch.as_tensor(np.concatenate(pos), device=device, dtype=torch.int64))
                data.append(torch.as_tensor(np.concatenate([g['data'][n] for n in names]), device=device))
            cross = {}
            for i in range(len(self.children)):
                for j in range(i + 1, len(self.children)):
                    blk = g['G'][np.ix_(g['blocks'][i], g['blocks'][j])]
                    m = np.zeros((blk.shape[0], pad(blk.shape[1])))
                    m[:, :blk.shape[1]] = blk
                    cross[(i, j)] = torch.as_tensor(m, device=device)
            g['dev'] = dict(idx=idx, data=data, cross=cross, pad=pad, bufs=None)
        return g['dev']

    def _eval_global_device(self, theta, out, status=None, models=None):
        """chi2 with a global covariance for CUDA walkers: every engine evaluates model and its diagonal term (its own
        stream, ordered by events as in `eval_device_tensor`), then the cross terms 2 r_i^T G_ij r_j - the product G_ij r_j on
        the first engine's product kernels (`vmx_matvec_device`), residuals and row sums as tensor glue."""
        import torch
        st = self._global_state(theta.device)
        B, n_eng = theta.shape[0], len(self.children)
        cap = max(B, self.max_batch)
        if st['bufs'] is None or st['bufs']['chi2'].shape[1] < B:
            st['bufs'] = dict(
                chi2=torch.empty((n_eng, cap), dtype=torch.float64, device=theta.device),
                status=torch.zeros((n_eng, cap), dtype=torch.int32, device=theta.device),
                model=[torch.empty((cap, c.model_size), dtype=torch.float64, device=theta.device) for c in self.children],
                res=[torch.zeros((cap, st['pad'](i.numel())), dtype=torch.float64, device=theta.device) for i in st['idx']],
                z=[torch.empty((cap, st['pad'](i.numel())), dtype=torch.float64, device=theta.device) for i in st['idx']])
        bufs = st['bufs']
        mine = torch.cuda.current_stream(theta.device)
        ready = mine.record_event()
        done = []
        for ci, c in enumerate(self.children):
            stream = self._stream_of(c.stream_handle(), theta.device)
            stream.wait_event(ready)
            c.eval_device(theta.data_ptr(), B, bufs['chi2'][ci].data_ptr(), bufs['model'][ci].data_ptr(), bufs['status'][ci].data_ptr())
            done.append(self._stream_of(c.last_stream_handle(), theta.device).record_event())
        for ev in done:
            mine.wait_event(ev)
        total = bufs['chi2'][:, :B].sum(dim=0)
        rows = None
        g = self._global
        if g.get('mock_index') is not None:
            # Monte-Carlo fits in lock-step: walker b is fitted to row mock_index[b] of every item's mock pool (< 0: the data)
            if g.get('pool_dev') is None:
                g['pool_dev'] = [torch.as_tensor(np.concatenate([g['pools'][n] for n in names], axis=1), device=theta.device)
                                 for names in self.groups]
            rows = torch.as_tensor(g['mock_index'][:B], device=theta.device)
        for ci in range(n_eng):
            n = st['idx'][ci].numel()
            data = st['data'][ci][None, :]
            if rows is not None:
                data = torch.where((rows >= 0)[:, None], g['pool_dev'][ci].index_select(0, rows.clamp(min=0)), data)
            bufs['res'][ci][:B, :n] = data - bufs['model'][ci][:B].index_select(1, st['idx'][ci])
        first = self.children[0]
        prod = self._stream_of(first.stream_handle(), theta.device)
        for (i, j), blk in st['cross'].items():
            prod.wait_event(mine.record_event())                    # the residuals are formed
            first.matvec_device(blk.data_ptr(), blk.shape[0], blk.shape[1], bufs['res'][j].data_ptr(), B, bufs['z'][i].data_ptr())
            mine.wait_event(prod.record_event())
            n = st['idx'][i].numel()
            total = total + 2.0 * (bufs['res'][i][:B, :n] * bufs['z'][i][:B, :n]).sum(dim=1)
        bad = bufs['status'][:, :B].max(dim=0).values != 0
        out.copy_(torch.where(bad, torch.full_like(total, SENTINEL), total))
        if status is not None:
            merged = bufs['status'][0, :B].clone()
            for ci in range(1, n_eng):
                merged |= bufs['status'][ci, :B]
            status.copy_(merged)
        if models is not None:
            for ci, src, dst in self._gather:
                models[:, dst] = bufs['model'][ci][:B, src]

    def _eval_global_host(self, theta, want_model):
        import torch
        dev = torch.device('cuda', self.children[0].device)
        with torch.cuda.device(dev):
            t = torch.as_tensor(theta, device=dev)
            B = t.shape[0]
            out = torch.empty(B, dtype=torch.float64, device=dev)
            status = torch.empty(B, dtype=torch.int32, device=dev)
            models = torch.empty((B, self.model_size), dtype=torch.float64, device=dev) if want_model else None
            self._eval_global_device(t, out, status, models)
            torch.cuda.current_stream(dev).synchronize()
        return out.cpu().numpy(), status.cpu().numpy(), (models.cpu().numpy() if want_model else None)

    def eval_device_tensor(self, theta, out):
        """``theta`` CUDA float64 [B, n_params] -> ``out`` CUDA float64 [B]: every engine evaluates the walkers on its own
        stream into its own buffer, ordered after the caller's stream by an event; the sum is formed on the caller's stream
        behind one event per engine - no host synchronisation anywhere."""
        import torch
        if self._global is not None:
            return self._eval_global_device(theta, out)
        B = theta.shape[0]
        if self._dev_bufs is None or self._dev_bufs.shape[1] < B:
            self._dev_bufs = torch.empty((len(self.children), max(B, self.max_batch)), dtype=torch.float64, device=theta.device)
        mine = torch.cuda.current_stream(theta.device)
        ready = mine.record_event()                 # theta (and the previous reader of the buffers) are done by then
        done = []
        for ci, c in enumerate(self.children):
            stream = self._stream_of(c.stream_handle(), theta.device)
            stream.wait_event(ready)
            c.eval_device(theta.data_ptr(), B, self._dev_bufs[ci].data_ptr())
            done.append(self._stream_of(c.last_stream_handle(), theta.device).record_event())
        for ev in done:
            mine.wait_event(ev)
        parts = self._dev_bufs[:, :B]
        total = parts.sum(dim=0)
        out.copy_(torch.where(parts.max(dim=0).values >= SENTINEL, torch.full_like(total, SENTINEL), total))

    def _stream_of(self, handle, device):
        import torch
        if not hasattr(self, '_ext_streams'):
            self._ext_streams = {}
        if handle not in self._ext_streams:
            self._ext_streams[handle] = torch.cuda.ExternalStream(handle, device=device)
        return self._ext_streams[handle]

    def eval_device(self, *args, **kwargs):
        raise NotImplementedError('an engine group evaluates device walkers through eval_device_tensor')

    def sync(self):
        for c in self.children:
            c.sync()

    def stream_handle(self):
        return self.children[0].stream_handle()

    def last_stream_handle(self):
        return self.children[-1].last_stream_handle()

    def set_lanes(self, lanes):
        if int(lanes) != 1:
            raise NotImplementedError('an engine group keeps one batch in flight per engine')

    # ---- per item
    def set_data(self, name, masked_data):
        self._owner[name].set_data(name, masked_data)
        if self._global is not None:
            self._global['data'][name] = np.asarray(masked_data, dtype=np.float64).copy()
        if self._global is not None and self._global['dev'] is not None:
            # (the cross terms' copy of the engine's masked data, in the engine's item order)
            import torch
            ci = self.children.index(self._owner[name])
            off = 0
            for n in self.groups[ci]:
                size = self.prob.items[n].data_size
                if n == name:
                    d = self._global['dev']['data'][ci]
                    d[off:off + size] = torch.as_tensor(np.asarray(masked_data, dtype=np.float64), device=d.device)
                off += size

    def set_mock_pool(self, name, pool):
        self._owner[name].set_mock_pool(name, pool)
        if self._global is not None:
            # (the cross terms read the walker's pool row of every item: kept per item, joined per engine on first use)
            self._global.setdefault('pools', {})[name] = np.atleast_2d(np.asarray(pool, dtype=np.float64)).copy()
            self._global['pool_dev'] = None

    def set_invcov(self, name, invcov):
        self._owner[name].set_invcov(name, invcov)

    def marg_coeff(self, name, B):
        return self._owner[name].marg_coeff(name, B)

    def metal_xi(self, item_name, pair_index):
        return self._owner[item_name].metal_xi(item_name, pair_index)

    # ---- every engine
    def _all(self, method, *args, **kwargs):
        return [getattr(c, method)(*args, **kwargs) for c in self.children]

    def set_mock_index(self, index=None):
        self._all('set_mock_index', index)
        if self._global is not None:
            self._global['mock_index'] = None if index is None else np.ascontiguousarray(index, dtype=np.int64)

    def set_constant_nl_hint(self, on=True, gaussian=False):
        self._all('set_constant_nl_hint', on, gaussian)

    def set_direct_pk(self, pk=None):
        self._all('set_direct_pk', pk)

    def set_quadratic_form(self, on=True):
        return all(self._all('set_quadratic_form', on))

    def set_mu_quadrature(self, node_rule=True):
        self._all('set_mu_quadrature', node_rule)

    def set_linear_spectra(self, pk_full, pk_smooth):
        self._all('set_linear_spectra', pk_full, pk_smooth)

    def set_parameter_transform(self, scale=None, shift=None):
        self._all('set_parameter_transform', scale, shift)

    def set_metal_beta_override(self, beta=None):
        self._all('set_metal_beta_override', beta)

    def set_profiling(self, on):
        self._all('set_profiling', on)

    def set_profiling_classes(self, names, stride=1):
        self._all('set_profiling_classes', names, stride)

    def timings(self, reset=True):
        total = {}
        for t in self._all('timings', reset):
            for k, (ms, n) in t.items():
                a = total.get(k, (0.0, 0))
                total[k] = (a[0] + ms, a[1] + n)
        return total

    def pk_multipoles(self, B=1):
        """dict (engine index, pipeline id) -> [B, 4, nk], the keys ``pipe_index`` holds."""
        return {(ci, pid): v for ci, c in enumerate(self.children) for pid, v in c.pk_multipoles(B).items()}

    def mu_nodes(self):
        return self.children[0].mu_nodes()

    # ---- products on the first engine's kernels
    def matvec_device(self, *args):
        self.children[0].matvec_device(*args)

    def matmul_host(self, A, X):
        return self.children[0].matmul_host(A, X)

    def debug_read(self, *args, **kwargs):
        raise NotImplementedError('stage taps are per engine: read them from EngineGroup.children')

    def close(self):
        for c in self.children:
            c.close()
        self.children = []
