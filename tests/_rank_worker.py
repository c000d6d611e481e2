# one rank of the multi-process GPU tests (not a test itself): python tests/_rank_worker.py
# env: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run sets them.
# Gathers a sharded right operand (RCCL over xGMI, or host-staged if RCCL is unavailable), checks it against the shards
# every rank can regenerate from the seeds, then checks this rank's block of commutes_termwise and of the product against
# the oracle.  Prints RANK_OK <rank> <data plane>.
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import _lib, kernels, parallel
from symmer_amd.kernels import DeviceOp
from oracle import oracle_c as oc

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
_lib.init(int(os.environ.get('LOCAL_RANK', '0')))
comm = parallel.Communicator.from_env()
want = os.environ.get('EXPECT_PLANE')
assert comm.gathers and (want is None or comm.data_plane == want), (comm.data_plane, comm.rccl_error)
n, M, N = 130, 1003, 77 * world + 5
wq = (n + 63) // 64
ts, bounds = parallel.shard_bounds(M, world)
mine = bounds[rank][1] - bounds[rank][0]
shard = parallel.padded_random_shard(mine, ts, n, 4242 + rank)
full = DeviceOp.alloc(ts * world, wq, with_coeff=True)
for _ in range(2):                                            # twice: the gather must be repeatable into the same handle
    comm.allgather_op(shard, full, M)
assert comm.verify_allgather(shard, full, M) and comm.data_plane == (want or comm.data_plane)     # the self-check agrees with a healthy plane
rows, coeff = full.download()
exp_r, exp_c = [], []
for r in range(world):
    o = DeviceOp.random(ts, n, 0.3, 4242 + r)
    er, ec = o.download(); o.free()
    k = bounds[r][1] - bounds[r][0]
    er[k:] = 0; ec[k:] = 0                                    # the padding of a short shard is zero rows
    exp_r.append(er); exp_c.append(ec)
exp_r, exp_c = np.vstack(exp_r)[:M], np.hstack(exp_c)[:M]
assert np.array_equal(rows, exp_r) and np.array_equal(coeff, exp_c), 'gathered operand differs from the shards'
# this rank's block of the left axis against the gathered right operand
left = DeviceOp.random(N, n, 0.3, 99)
lr, lc = left.download()
_, lb = parallel.shard_bounds(N, world)
b0, b1 = lb[rank]
assert np.array_equal(kernels.commutes(lr[b0:b1], rows), oc.commutes(lr[b0:b1], exp_r)), 'commutation block mismatch'
pr, pc = kernels.mul_allpairs(lr[b0:b1], lc[b0:b1], rows, coeff, True)
er, ec = oc.mul_allpairs(lr[b0:b1], lc[b0:b1], exp_r, exp_c, True)
assert np.array_equal(pr, er) and np.array_equal(pc, ec), 'product block mismatch'      # same un-fused IEEE expression on both sides
# product + cleanup with the OUTER operand sharded (SURVEY 8e stretch): rows, row order and coefficients of the single-process oracle
rng = np.random.default_rng(17)
n2, Ni2, No2 = 70, 300, 201
from oracle import oracle_np as onp
A2 = onp.pack_rows(rng.random((Ni2, 2 * n2)) < 0.4); B2 = onp.pack_rows(rng.random((No2, 2 * n2)) < 0.4)
A2[50:80] = A2[:30]; B2[100:120] = B2[:20]                    # duplicate rows: merges inside and across the ranks' blocks
a2 = (rng.integers(-8, 9, Ni2) + 1j * rng.integers(-8, 9, Ni2)) / 16.0; b2 = (rng.integers(-8, 9, No2) + 1j * rng.integers(-8, 9, No2)) / 16.0
_, ob = parallel.shard_bounds(No2, world)
inner = DeviceOp.upload(A2, a2)
outer = DeviceOp.upload(B2[ob[rank][0]:ob[rank][1]], b2[ob[rank][0]:ob[rank][1]])
res = comm.mul_cleanup_sharded(inner, outer, True, 1e-15)
rr, rc = res.download()
er, ec = oc.mul(A2, a2, B2, b2)
assert np.array_equal(rr, er) and np.array_equal(rc, ec), 'sharded product + cleanup differs from the oracle'
for h in (inner, outer, res):
    h.free()
# the same product, and the squared operator, with the PAIRS partitioned by the linear class of their product row (both operands complete on
# every rank, the share computed / merged on the device, shares all-gathered and ordered by pair index): the replicated result is the oracle's
innerf = DeviceOp.upload(A2, a2); outerf = DeviceOp.upload(B2, b2)
for X, Y, ex in ((innerf, outerf, (A2, a2, B2, b2)), (innerf, innerf, (A2, a2, A2, a2))):
    st = {}
    res = comm.mul_cleanup_hash_partitioned(X, Y, True, 1e-15, stats=st)
    rr, rc = res.download()
    pr, pc = oc.mul_allpairs(*ex, True)
    er, ec = oc.cleanup(pr, pc, 1e-15)
    assert np.array_equal(rr, er) and np.array_equal(rc, ec), 'hash-partitioned product + cleanup differs from the oracle'
    assert st['keys_exchanged'] == 0 and 0 < st['pairs_owned'] < st['pairs_total']
    res.free()
innerf.free(); outerf.free()
comm.barrier()
plane = comm.data_plane
comm.close()
print(f'RANK_OK {rank} {plane}', flush=True)
