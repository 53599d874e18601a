for e in 0 1 2 4 7; do echo "=== RGM_G144_EXP=$e"; RGM_G144_EXP=$e python tools/g144_stamp.py 4096 1152 1152 2>&1 | grep -v amdgpu.ids; done
