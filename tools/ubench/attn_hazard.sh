#!/bin/bash
# builds the three flavours of the short-sequence attention hazard probe (see attn_hazard.hip) next to this script
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -DRGM_ATTN_HAZARD_DBG -I../../rule-guided-music_amd/csrc"
# round 5: the two-phase Q prologue is the product kernel; -DRGM_ATTN_ONE_PHASE_Q restores the interleaved prologue the hazard was found in
/opt/rocm/bin/hipcc $F -DRGM_ATTN_ONE_PHASE_Q -o attn_hazard attn_hazard.hip
/opt/rocm/bin/hipcc $F -o attn_hazard_two_phase attn_hazard.hip
/opt/rocm/bin/hipcc $F -DRGM_ATTN_ONE_PHASE_Q -DRGM_ATTN_HAZARD_DUMP -o attn_hazard_dump attn_hazard.hip
