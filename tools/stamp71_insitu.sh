export RGM_LIB_PATH=$PWD/rule-guided-music_amd/rgm/librgm_hip_stamp.so
python tools/gemm2_insitu_stamp.py 16 2>&1 | grep -v amdgpu
python tools/gemm_stamp.py 4096 4096 1152 71 2>&1 | grep -v amdgpu
