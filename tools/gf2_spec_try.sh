export TMPDIR=/tmp
timeout 900 python3 tests/stress_gf2.py 3 100 2>&1 | tail -4
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "rref or gf2 or symgen or symmetry or cfg4 or elimination or independent" 2>&1 | tail -3
bash tools/gf2_quick.sh
