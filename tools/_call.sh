PLANER_HIP_LIB=$PWD/planer_amd/build/ab/libga_stamp.so python tools/wf4_stamp.py 2>&1 | tee gpurun_out/r6h/wf4_stamps_ga.txt
