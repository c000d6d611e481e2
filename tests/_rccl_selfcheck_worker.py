# one rank, SYMGPU_FORCE_COMM=1: the RCCL data plane's self-check (Communicator.verify_allgather).  A healthy gather passes; a
# gathered operand that was corrupted afterwards is detected, RCCL is left for the host-staged plane and the operand is repaired.
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symmer_amd import _lib, parallel
from symmer_amd.kernels import DeviceOp
_lib.init(0)
comm = parallel.Communicator.from_env()
assert comm.gathers and comm.data_plane == 'rccl', (comm.data_plane, comm.rccl_error)
n, M = 300, 900
ts, _ = parallel.shard_bounds(M, 1)
shard = parallel.padded_random_shard(M, ts, n, 11)
full = DeviceOp.alloc(ts, (n + 63) // 64, with_coeff=True)
comm.allgather_op(shard, full, M)
assert comm.verify_allgather(shard, full, M) and comm.degraded is None
good_r, good_c = full.download()
bad = good_r[5:6].copy(); bad[0, 0] ^= 1 << 17
_lib.check(_lib.lib().symgpu_op_write(full.handle, 5, bad.ctypes.data, None, 1))           # one flipped bit in row 5
full.set_rows(M)
assert not comm.verify_allgather(shard, full, M)
assert comm.data_plane == 'host-staged' and 'differ' in comm.degraded, (comm.data_plane, comm.degraded)
r, c = full.download()
assert np.array_equal(r, good_r) and np.array_equal(c, good_c), 'the fallback must have repaired the operand'
comm.allgather_op(shard, full, M)                                                            # and keeps working
r, c = full.download()
assert np.array_equal(r, good_r) and np.array_equal(c, good_c)
comm.close()
print('SELFCHECK_OK', flush=True)
