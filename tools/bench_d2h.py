# ad-hoc: library download rate into fresh / touched numpy memory
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from symmer_amd import _lib, kernels
lib = _lib.lib()
nbytes = 3 << 30
buf = ctypes.c_void_p(); _lib.check(lib.symgpu_dev_alloc(nbytes, ctypes.byref(buf)))
for trial in range(3):
    a = np.empty(nbytes // 8, dtype=np.uint64)
    t0 = time.perf_counter(); _lib.check(lib.symgpu_dev_download(buf, a.ctypes.data, nbytes)); t1 = time.perf_counter()
    _lib.check(lib.symgpu_dev_download(buf, a.ctypes.data, nbytes)); t2 = time.perf_counter()
    print(f'fresh np.empty: {nbytes/(t1-t0)/1e9:.1f} GB/s   same array again: {nbytes/(t2-t1)/1e9:.1f} GB/s', flush=True)
    del a
