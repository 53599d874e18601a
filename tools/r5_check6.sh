for v in 0 2 0 2; do
export RGM_ST_PLAIN=$v
echo "=== RGM_ST_PLAIN=$v"
python tools/g144_insitu_stamp.py 16 2>&1 | grep -v "amdgpu\|consumer [123]\|loader   [567]"
done
