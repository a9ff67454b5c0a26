mkdir -p gpurun_out/r6k
PLANER_HIP_LIB=$PWD/planer_amd/build/ab/libhalf_stamp.so python tools/wf4_stamp.py 2>&1 | tee gpurun_out/r6k/wf4_stamps_half.txt | grep -v "wave [4-7]" 
