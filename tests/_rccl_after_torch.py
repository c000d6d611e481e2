# torch imported (and its CUDA runtime initialised) BEFORE the library: RCCL init + all-gather must still work
import torch, ctypes, sys, os
print('torch cuda available:', torch.cuda.is_available(), flush=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import _lib, packing
from symmer_amd.kernels import DeviceOp
lib = _lib.lib()
ident = (ctypes.c_uint8 * 128)()
_lib.check(lib.symgpu_comm_unique_id(ctypes.addressof(ident)))
_lib.check(lib.symgpu_comm_init(ctypes.addressof(ident), 0, 1))
rows = packing.pack_rows(np.random.default_rng(0).random((100, 260)) < 0.3)
shard = DeviceOp.upload(rows, np.ones(100, dtype=complex)); full = DeviceOp.alloc(100, rows.shape[1] // 2, True)
_lib.check(lib.symgpu_comm_allgather_op(shard.handle, full.handle))
assert np.array_equal(full.download()[0], rows)
_lib.check(lib.symgpu_comm_destroy())
print('rccl after torch OK', flush=True)
