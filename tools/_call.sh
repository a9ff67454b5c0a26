mkdir -p gpurun_out/r6m
PLANER_HIP_LIB=$PWD/planer_amd/build/ab/libdma0.so python -m pytest tests/test_gpu_wf4.py -x -q 2>&1 | tail -2
bash tools/ab_lib.sh planer_amd/build/ab/libdma0.so 3 2>&1 | tail -12 | tee gpurun_out/r6m/ab_dma0.txt
