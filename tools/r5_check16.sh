python -m pytest tests/test_gpu_fullsize.py -x -q -k "tile_144 or heuristic_decompositions" 2>&1 | tail -2
bash tools/ab_lib.sh rule-guided-music_amd/rgm/librgm_hip_prev.so --steps 20 --warmup 5
bash tools/ab_lib.sh rule-guided-music_amd/rgm/librgm_hip_prev.so --batch 4 --steps 20 --warmup 5
python tools/g144_stamp.py 4096 1152 1152 2>&1 | grep -v "amdgpu\|consumer [123]\|loader   [567]"
