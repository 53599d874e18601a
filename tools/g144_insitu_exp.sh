#!/bin/bash
# elimination runs of the 128x144 kernel IN the forward (timing only, wrong results): RGM_G144_EXP 1 no barrier, 2 no fragment reads, 4 no LDS-DMA, 7 all
for t in 15 11; do for e in 0 1 2 4 7; do echo "=== RGM_T144=$t RGM_G144_EXP=$e"; RGM_T144=$t RGM_G144_EXP=$e timeout 300 python tools/g144_insitu_stamp.py 16 2>&1 | grep -v "amdgpu\|consumer [123]\|loader   [567]"; done; done
