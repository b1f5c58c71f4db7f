"""Partition of the correlation items by transform setting (no GPU: the partition is host logic)."""
from types import SimpleNamespace as NS

import pytest

from vega_amd.engine_group import item_settings, setting_groups


def _item(n_mu=1000, old=False, lowring=True, metals=(), fvoigt=None):
    pipe = lambda: NS(pk=NS(n_mu=n_mu, fvoigt_table=fvoigt), xi=NS(old_fftlog=old, fht_lowring=lowring))
    return NS(core=pipe(), metals=[NS(pipeline=pipe()) for _ in metals])


def test_items_are_grouped_by_setting_in_configured_order():
    prob = NS(items={'a': _item(), 'b': _item(lowring=False), 'c': _item(metals=(1, 2)), 'd': _item(n_mu=400, lowring=False),
                     'e': _item(lowring=False)})
    assert setting_groups(prob) == [['a', 'c'], ['b', 'e'], ['d']]
    assert setting_groups(NS(items={'a': _item(), 'c': _item()})) == [['a', 'c']]


def test_voigt_tables_split_items_and_table_free_items_join_a_group():
    import numpy as np
    t1, t2 = np.arange(6.).reshape(3, 2), np.arange(6.).reshape(3, 2) + 1
    prob = NS(items={'a': _item(), 'b': _item(fvoigt=t1), 'c': _item(fvoigt=t2), 'd': _item(fvoigt=t1.copy()),
                     'e': _item(lowring=False)})
    assert setting_groups(prob) == [['a', 'b', 'd'], ['c'], ['e']]


def test_disagreement_inside_one_item_is_an_error():
    item = _item(metals=(1,))
    item.metals[0].pipeline.xi.old_fftlog = True
    with pytest.raises(ValueError, match='disagree'):
        item_settings(item)
