import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: soaks and repeated sweeps of opt-in modes; deselected unless --runslow / ATACOM_SLOW=1")
    config.addinivalue_line("markers", "mapping(name=, dt=, kind=): what a lanes-parametrised test runs when it has no such parameter")


def pytest_addoption(parser):
    parser.addoption('--runslow', action='store_true', default=False, help='also run the tests marked slow')


def mapping_exists(name, dt, lanes, kind='step', dyn=False):
    """The census of kernel mappings (lanes per environment) the library instantiates -- a mirror of
    rl_on_manifold_amd/csrc/atacom_ops_impl.h: has_mapping, held against the library by
    tests/test_gpu_rollout.py::test_mapping_census_matches_the_library.  kind: 'step' (= the T-step kernels) or 'mlp'."""
    if lanes == 1:
        return True
    if name.startswith('circle'):
        return False
    if dyn:
        return lanes == 4
    if dt == 'f64':
        return lanes == 4 if kind == 'mlp' else (lanes == 4 or (lanes == 8 and name == 'iiwa'))
    return lanes in (2, 4, 8)


def pytest_collection_modifyitems(config, items):
    """(1) A test parametrised over `lanes` only runs the mappings that exist for its (environment, dtype): the library would
    narrow the request to the next instantiated mapping and the case would repeat another one.  The environment / dtype come
    from the test's own `name` / `dt` parameters or from @pytest.mark.mapping(...).  (2) slow tests need --runslow."""
    runslow = config.getoption('--runslow') or os.environ.get('ATACOM_SLOW') == '1'
    keep, drop = [], []
    for it in items:
        cs = getattr(it, 'callspec', None)
        if it.get_closest_marker('slow') is not None and not runslow:
            drop.append(it)
            continue
        if cs is not None and 'lanes' in cs.params:
            m = it.get_closest_marker('mapping')
            d = dict(m.kwargs) if m else {}
            name = cs.params.get('name', d.get('name'))
            dt = cs.params.get('dt', d.get('dt', 'f32'))
            if name is not None and not mapping_exists(name, dt, cs.params['lanes'], d.get('kind', 'step'), d.get('dyn', False)):
                drop.append(it)
                continue
        keep.append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load


@pytest.fixture(scope='session')
def lib_built():
    """Build the HIP library once per session (hipcc cross-compiles gfx950 without a GPU)."""
    from rl_on_manifold_amd import build
    return build.build(verbose=False)
