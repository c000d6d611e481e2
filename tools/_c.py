import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from symmer_amd import kernels
from symmer_amd.kernels import DeviceOp
N, n = int(sys.argv[1]), int(sys.argv[2])
A = DeviceOp.random(N, n, 0.3, seed=3)
for _ in range(5):
    r = kernels.mul_cleanup_handles(A, A, True, 1e-15); kernels.sync(); r.free()
import time; time.sleep(0.01)
r = kernels.mul_cleanup_handles(A, A, True, 1e-15); kernels.sync()
