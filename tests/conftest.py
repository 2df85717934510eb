import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this environment')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def rel_err(got, want):
    """Normwise relative error max|got - want| / max|want| (the parity metric for features)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    if want.size == 0:
        return 0.0
    return float(np.max(np.abs(got - want)) / max(np.max(np.abs(want)), 1e-30))


def elementwise_err(got, want, floor=0.1):
    """Elementwise relative error with an absolute floor: max over entries of
    |got - want| / max(|want|, floor * max|want|).  An entry of a float32 matrix product carries an
    absolute rounding error set by the size of the terms that were summed, not by its own size, so
    entries far below the tensor's scale cannot be held to a relative bound; `floor` says from which
    fraction of the scale downwards the bound turns absolute."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    if want.size == 0:
        return 0.0
    top = max(float(np.max(np.abs(want))), 1e-30)
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), floor * top)))


def assert_close(got, want, tol=1e-5, what=''):
    """The float32 parity gates: normwise max|d| / max|want| <= tol; elementwise
    |d| <= tol * max(|want|, 0.1 max|want|) (every entry within a decade of the scale holds `tol`
    relative to itself, smaller ones tol/10 of the scale absolutely); and, two decades down,
    |d| <= 10 tol * max(|want|, 0.01 max|want|).  The last one is as far as two float32 evaluations
    of the same dot products can be held to each other: with the floor at 0.01 the results here differ
    from the float32 oracle by 1e-5 .. 8e-5 elementwise, which is the oracle's own distance from the
    float64 value (assert_bracketed measures exactly that where a float64 run is affordable)."""
    n, e, e2 = rel_err(got, want), elementwise_err(got, want), elementwise_err(got, want, 0.01)
    assert n <= tol and e <= tol and e2 <= 10 * tol, \
        f'{what}: normwise {n:.2e}, elementwise {e:.2e} (gate {tol:.0e}), at the 0.01 floor {e2:.2e} (gate {10 * tol:.0e})'
    return n, e


def exact_inputs(x, graph, weights):
    """float64 copies of a forward's inputs: the oracle run on them is the exact value the float32
    results are bracketed against."""
    w64 = {k: np.asarray(v, np.float64) for k, v in weights.items()}
    g64 = {'edge_index': list(graph['edge_index']), 'edges': [np.asarray(e, np.float64) for e in graph['edges']],
           'node_loc': graph['node_loc'], 'global': np.asarray(graph['global'], np.float64)}
    return np.asarray(x, np.float64), g64, w64


def assert_bracketed(got, want32, exact, tol=1e-5, what=''):
    """The parity gate that does not depend on where an absolute floor is put: `exact` is the same
    computation in float64, `want32` the float32 oracle.  The result under test must be within `tol`
    of the exact value normwise, and no further from it than twice the float32 oracle is -- normwise,
    and elementwise with the floor three decades below the tensor's scale (where two float32
    evaluations of a long dot product no longer agree to 1e-5 with each other, only with the exact
    value to the same degree)."""
    hip, ora = rel_err(got, exact), rel_err(want32, exact)
    assert hip <= tol and hip <= 2 * ora + 1e-7, f'{what}: normwise {hip:.2e} (the float32 oracle: {ora:.2e})'
    hip_e, ora_e = elementwise_err(got, exact, 1e-3), elementwise_err(want32, exact, 1e-3)
    assert hip_e <= 2 * ora_e + 1e-6, f'{what}: elementwise {hip_e:.2e} (the float32 oracle: {ora_e:.2e})'
    return hip, ora, hip_e, ora_e


@pytest.fixture(scope='session')
def oracle():
    import gunet_oracle
    return gunet_oracle
