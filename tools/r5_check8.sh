export RGM_LIB_PATH=$PWD/rule-guided-music_amd/rgm/librgm_hip_stamp.so
python tools/conv_stamp.py 64 2>&1 | grep -v amdgpu
RGM_GEMM2_DBG_TILE=72 python tools/conv_stamp.py 64 2>&1 | grep -v amdgpu | head -6
