#!/bin/bash
# in-forward stamps of the gemm2 tiles a forward runs (stamped build: see tools/gemm2_insitu_stamp.py): the LAST launch of each tile id
export RGM_LIB_PATH=$PWD/rule-guided-music_amd/rgm/librgm_hip_stamp.so
echo "##### B = 16: 71 = 256x256 (fc1's main launch), 56 = 128x64 loader/consumer (fc1's last 512 columns)"
for t in 71 56; do RGM_GEMM2_DBG_TILE=$t python tools/gemm2_insitu_stamp.py 16 2>&1 | grep -v amdgpu; done
echo "##### B = 4: 54 = 128x128 loader/consumer (qkv), 56 = 128x64 (proj)"
for t in 54 56; do RGM_GEMM2_DBG_TILE=$t python tools/gemm2_insitu_stamp.py 4 2>&1 | grep -v amdgpu; done
echo "##### B = 2: 56 / 57"
for t in 56 57; do RGM_GEMM2_DBG_TILE=$t python tools/gemm2_insitu_stamp.py 2 2>&1 | grep -v amdgpu; done
