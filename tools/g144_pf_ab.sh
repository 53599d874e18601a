for pf in 0 1; do
  echo "##### RGM_G144_PF=$pf"
  export RGM_G144_PF=$pf
  COLD=1 python tools/g144_stamp.py 4096 1152 1152 1024 4608 1152 2>&1 | grep -v "amdgpu\|consumer [123]\|loader   [567]"
  RGM_T144=11 python tools/g144_insitu_stamp.py 16 2>&1 | grep -v "amdgpu\|consumer [123]\|loader   [567]"
  RGM_T144=15 python tools/g144_insitu_stamp.py 16 2>&1 | grep -v "amdgpu\|consumer [123]\|loader   [567]"
  RGM_T144=1 python tools/g144_insitu_stamp.py 4 2>&1 | grep -v "amdgpu\|consumer [123]\|loader   [567]"
  RGM_T144=9 python tools/g144_insitu_stamp.py 4 8 2>&1 | grep -v "amdgpu\|consumer [123]\|loader   [567]"
done
