for lib in "" planer_amd/build/ab/libga9.so planer_amd/build/ab/libga3.so "" planer_amd/build/ab/libga9.so; do echo "== ${lib:-shipped (6 slots, 5 ahead)}"; env ${lib:+PLANER_HIP_LIB=$PWD/$lib} python tools/wf4_bench.py --shapes 64x56,128x28 --algos 9 2>&1 | tail -2; done
PLANER_HIP_LIB=$PWD/planer_amd/build/ab/libga9.so python -m pytest tests/test_gpu_wf4.py -x -q 2>&1 | tail -2
