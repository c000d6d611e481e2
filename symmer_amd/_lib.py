"""ctypes binding of ``libsymgpu.so`` (C ABI declared in ``include/symgpu.h``).

There is NO CPU fallback: if the shared library is missing, or no MI355X is visible, every hot-path
call raises :class:`SymgpuError`.  The library is built in-tree by ``__graft_entry__.build()``
(``make -C symmer_amd/csrc``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libsymgpu.so')

OK, E_INVALID, E_HIP, E_NOMEM, E_CAPACITY, E_NODEVICE, E_COLLISION, E_RCCL = 0, -1, -2, -3, -4, -5, -6, -7
_NAMES = {E_INVALID: 'SYMGPU_E_INVALID', E_HIP: 'SYMGPU_E_HIP', E_NOMEM: 'SYMGPU_E_NOMEM',
          E_CAPACITY: 'SYMGPU_E_CAPACITY', E_NODEVICE: 'SYMGPU_E_NODEVICE', E_COLLISION: 'SYMGPU_E_COLLISION',
          E_RCCL: 'SYMGPU_E_RCCL'}


class SymgpuError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f'{_NAMES.get(code, code)}: {text}')
        self.code = code


c_i64, c_int, c_dbl, c_u64 = ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_uint64
P = ctypes.c_void_p          # every pointer argument is passed as an address
PP = ctypes.POINTER(ctypes.c_void_p)

# name -> argument types, exactly the prototypes of include/symgpu.h (all return int unless noted)
SIGNATURES = {
    'symgpu_init': [c_int],
    'symgpu_shutdown': [],
    'symgpu_init_all': [c_int],
    'symgpu_set_device': [c_int],
    'symgpu_current_device': [P],
    'symgpu_n_initialised': [P],
    'symgpu_device_count': [P],
    'symgpu_sync': [],
    'symgpu_device_sync': [],
    'symgpu_device_name': [P, c_int],
    'symgpu_mem_info': [P, P],
    'symgpu_timer_start': [],
    'symgpu_timer_stop': [P],
    'symgpu_prof_enable': [c_int, c_int],
    'symgpu_prof_read': [c_int, P, P],
    'symgpu_debug_counter': [c_int, P],
    'symgpu_debug_rotation_trace': [P, c_int, P],
    'symgpu_degraded': [P, c_int],
    'symgpu_membw_probe': [c_i64, P, P],
    'symgpu_op_upload': [P, P, c_i64, c_int, PP],
    'symgpu_op_alloc': [c_i64, c_int, c_int, PP],
    'symgpu_op_download': [P, P, P, c_i64],
    'symgpu_op_info': [P, P, P, P],
    'symgpu_op_free': [P],
    'symgpu_op_set_rows': [P, c_i64],
    'symgpu_op_write': [P, c_i64, P, P, c_i64],
    'symgpu_op_random': [c_i64, c_int, c_dbl, c_u64, PP],
    'symgpu_op_checksum': [P, P, P],
    'symgpu_op_clone': [P, PP],
    'symgpu_op_set_coeff': [P, P],
    'symgpu_op_scale': [P, c_dbl, c_dbl, c_int],
    'symgpu_op_ycount': [P, P],
    'symgpu_op_upload_bool': [P, P, c_i64, c_int, PP],
    'symgpu_op_download_bool': [P, c_int, P, c_i64],
    'symgpu_ycount': [P, c_i64, c_int, P],
    'symgpu_commutes': [P, c_i64, P, c_i64, c_int, P],
    'symgpu_commutes_dev': [P, c_i64, c_i64, P, P],
    'symgpu_commutes_bits_dev': [P, c_i64, c_i64, P, P],
    'symgpu_dev_alloc': [c_i64, PP],
    'symgpu_dev_free': [P],
    'symgpu_dev_download': [P, P, c_i64],
    'symgpu_dev_upload': [P, P, c_i64],
    'symgpu_dev_checksum_u8': [P, c_i64, P],
    'symgpu_dev_popcount_u64': [P, c_i64, P],
    'symgpu_op_popcount': [P, P],
    'symgpu_op_copy_rows': [P, c_i64, P, c_i64, c_i64],
    'symgpu_mul_allpairs': [P, P, c_i64, P, P, c_i64, c_int, c_int, P, P],
    'symgpu_mul_allpairs_dev': [P, P, c_i64, c_i64, c_int, P],
    'symgpu_cleanup': [P, P, c_i64, c_int, c_dbl, c_int, P, P, c_i64, P],
    'symgpu_cleanup_dev': [P, c_dbl, c_int, PP],
    'symgpu_mul_cleanup': [P, P, c_i64, P, P, c_i64, c_int, c_int, c_dbl, c_int, P, P, c_i64, P],
    'symgpu_mul_cleanup_dev': [P, P, c_int, c_dbl, c_int, PP],
    'symgpu_cleanup_indexed_dev': [P, c_dbl, c_int, PP],
    'symgpu_mul_cleanup_indexed_dev': [P, P, c_int, c_dbl, c_int, PP],
    'symgpu_op_first_index': [P, P, c_i64],
    'symgpu_op_gather': [P, P, c_i64, PP],
    'symgpu_part_global_index': [P, P, c_i64, P, c_i64, c_i64],
    'symgpu_op_set_first_index': [P, P],
    'symgpu_merge_indexed_dev': [P, c_int, c_int, c_int, c_dbl, c_int, PP],
    'symgpu_rotate_single': [P, P, c_i64, c_int, P, c_dbl, c_dbl, c_int, c_dbl, P, P, c_i64, P, P],
    'symgpu_rotate_single_dev': [P, P, c_dbl, c_dbl, c_int, c_dbl, PP, P],
    'symgpu_rotate_single_dev_n': [P, P, c_dbl, c_dbl, c_int, c_dbl, PP, P, P],
    'symgpu_rotate_clifford_chain_dev': [P, P, P, c_i64, PP],
    'symgpu_perform_rotations_dev': [P, P, P, P, P, c_i64, c_dbl, c_int, PP, P, P, P],
    'symgpu_rref': [P, c_i64, c_i64, P, P],
    'symgpu_rref_dev': [P, c_i64, c_i64, P, P],
    'symgpu_symmetry_kernel': [P, c_i64, c_int, c_int, P, c_i64, P, P],
    'symgpu_symmetry_kernel_dev': [P, c_int, P, c_i64, P, P],
    'symgpu_op_gf2_rank': [P, P],
    'symgpu_generators_dev': [P, PP],
    'symgpu_generator_reconstruction_dev': [P, P, c_int, P, P],
    'symgpu_project_dev': [P, P, c_int, P, P, c_int, c_int, c_dbl, c_int, PP, P],
    'symgpu_noncontextual_dev': [P, P],
    'symgpu_state_inner_dev': [P, P, P],
    'symgpu_comm_available': [],
    'symgpu_comm_unique_id': [P],
    'symgpu_comm_init': [P, c_int, c_int],
    'symgpu_comm_destroy': [],
    'symgpu_comm_abandon': [],
    'symgpu_comm_allgather_op': [P, P],
    'symgpu_comm_barrier': [],
    'symgpu_comm_init_all': [c_int],
    'symgpu_comm_allgather_ops': [P, P, c_int],
}

_lib = None
_initialised_device = None


def load():
    """dlopen libsymgpu.so and attach prototypes (no GPU call)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SymgpuError(E_NODEVICE, f'{LIB_PATH} not found — build it with `python -c "import __graft_entry__ as g; g.build()"` '
                                      f'or `make -C symmer_amd/csrc` (there is no CPU fallback)')
    # RTLD_DEEPBIND: bind HIP symbols to the runtime this library was linked against (/opt/rocm), even if another HIP
    # runtime (e.g. the one bundled in a PyTorch wheel) is already loaded in the process' global scope.
    lib = ctypes.CDLL(LIB_PATH, mode=os.RTLD_NOW | os.RTLD_LOCAL | getattr(os, 'RTLD_DEEPBIND', 0))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.symgpu_last_error.restype = ctypes.c_char_p
    lib.symgpu_last_error.argtypes = []
    _lib = lib
    return lib


def last_error():
    return load().symgpu_last_error().decode('utf-8', 'replace')


def check(rc):
    if rc != OK:
        raise SymgpuError(rc, last_error())


def degraded():
    """Fast paths the library has switched off in this process (in-kernel wait timed out / attribute refused), as a list of sentences."""
    buf = ctypes.create_string_buffer(2048)
    check(load().symgpu_degraded(ctypes.addressof(buf), 2048))
    text = buf.value.decode('utf-8', 'replace')
    return [t for t in text.split('; ') if t]


def device_count():
    n = c_int(0)
    check(load().symgpu_device_count(ctypes.addressof(n)))
    return n.value


def init(device=None):
    """Initialise the context on ``device`` (default: $LOCAL_RANK or 0).  Raises if no GPU."""
    global _initialised_device
    lib = load()
    if device is None:
        if _initialised_device is not None:
            return _initialised_device
        device = int(os.environ.get('SYMGPU_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    if _initialised_device == device:
        return device
    n = device_count()
    if n <= 0:
        raise SymgpuError(E_NODEVICE, 'no HIP device visible: the symplectic hot path needs an MI355X (no CPU fallback)')
    check(lib.symgpu_init(device % n))
    _initialised_device = device % n
    return _initialised_device


def set_device(device):
    """Make ``device`` the calling thread's current device (its context is created on first use)."""
    check(load().symgpu_set_device(int(device)))


def current_device():
    d = c_int(-1)
    check(load().symgpu_current_device(ctypes.addressof(d)))
    return d.value


def lib():
    """The loaded library with an initialised context."""
    if _initialised_device is None:
        init()
    return _lib


def addr(a):
    """Address of a C-contiguous numpy array (or None)."""
    return None if a is None else a.ctypes.data
