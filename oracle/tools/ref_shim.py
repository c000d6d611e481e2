"""Import shim that lets the *reference* package be imported in the BUILD container only.

TEST INFRASTRUCTURE — never imported by the product, never shipped to the GPU box with
reference files (``/root/reference`` does not exist there).  It is used by
``oracle/tools/gen_golden.py`` (fixture generation) and ``oracle/tools/check_oracle_vs_ref.py``
(oracle pinning) and nothing else.

The reference needs qiskit / numba / openfermion / cached_property / ray / quimb ..., none of which
is installed here and there is no network.  The shim
  (1) answers every import under those names with a MagicMock module,
  (2) maps ``numba.njit`` to the identity decorator and ``prange`` to ``range``,
  (3) maps ``cached_property`` to ``functools.cached_property``,
  (4) provides ``qiskit._accelerate.sparse_pauli_op.unordered_unique`` with the qiskit-1.2.4
      semantics (first-occurrence indices + inverse map; restated from the published Rust:
      iterate rows, HashMap<row,id>, push i on first sight, inverses[i]=id).
Use:  ``import ref_shim`` BEFORE ``import symmer``.
"""
import sys, types, functools, importlib.abc, importlib.machinery
from unittest import mock
import numpy as np

REFERENCE_ROOT = '/root/reference'
MISSING = ('qiskit', 'numba', 'openfermion', 'cached_property', 'ray', 'quimb', 'ncon',
           'opt_einsum', 'cotengra')


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split('.')[0] in MISSING:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__name__ = spec.name; m.__path__ = []; m.__spec__ = spec; m.__loader__ = self
        return m

    def exec_module(self, module):
        pass


def unordered_unique(arr):
    a = np.ascontiguousarray(arr)
    v = a.view(np.dtype((np.void, a.dtype.itemsize * a.shape[1]))).ravel()
    _, first, inv = np.unique(v, return_index=True, return_inverse=True)
    order = np.argsort(first, kind='stable')
    rank = np.empty_like(order); rank[order] = np.arange(order.size)
    return first[order], rank[inv.ravel()]


def install():
    if getattr(sys, '_symmer_ref_shim', False):
        return
    sys._symmer_ref_shim = True
    sys.meta_path.insert(0, _Finder())
    nb = types.ModuleType('numba'); nb.__path__ = []
    nb.njit = lambda *a, **k: a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f)
    nb.prange = range
    errs = types.ModuleType('numba.core.errors')
    errs.NumbaDeprecationWarning = type('NumbaDeprecationWarning', (Warning,), {})
    errs.NumbaPendingDeprecationWarning = type('NumbaPendingDeprecationWarning', (Warning,), {})
    core = types.ModuleType('numba.core'); core.__path__ = []
    cp = types.ModuleType('cached_property'); cp.cached_property = functools.cached_property
    acc = types.ModuleType('qiskit._accelerate.sparse_pauli_op'); acc.unordered_unique = unordered_unique
    q = types.ModuleType('qiskit'); q.__path__ = []
    qa = types.ModuleType('qiskit._accelerate'); qa.__path__ = []
    qi = types.ModuleType('qiskit.quantum_info'); qi.SparsePauliOp = type('SparsePauliOp', (), {})
    of = types.ModuleType('openfermion'); of.QubitOperator = type('QubitOperator', (), {})
    of.count_qubits = lambda *a: 0
    for m in (q, qa, qi, of, acc):
        m.__getattr__ = (lambda name, _n=m.__name__: mock.MagicMock(name=_n + '.' + name))
    sys.modules.update({'numba': nb, 'numba.core': core, 'numba.core.errors': errs,
                        'cached_property': cp, 'qiskit': q, 'qiskit._accelerate': qa,
                        'qiskit._accelerate.sparse_pauli_op': acc, 'qiskit.quantum_info': qi,
                        'openfermion': of})
    sys.path.insert(0, REFERENCE_ROOT)


install()
