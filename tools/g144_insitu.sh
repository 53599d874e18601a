RGM_T144=11 python tools/g144_insitu_stamp.py 16 2>&1 | grep -v amdgpu
RGM_T144=15 python tools/g144_insitu_stamp.py 16 2>&1 | grep -v amdgpu
RGM_T144=1 python tools/g144_insitu_stamp.py 4 2>&1 | grep -v amdgpu
RGM_T144=9 python tools/g144_insitu_stamp.py 4 2>&1 | grep -v amdgpu
