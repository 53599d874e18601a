export RGM_LIB_PATH=$PWD/rule-guided-music_amd/rgm/librgm_hip_stamp.so
for cfg in "0 0" "2 0" "0 1" "2 1"; do set -- $cfg; echo "=== ACT=$1 SPLIT=$2"; ACT=$1 SPLIT=$2 python tools/gemm_stamp.py 4096 4096 1152 71 2>&1 | grep -v amdgpu | grep "tile 71\|   0 \|epilogue wave 0"; done
