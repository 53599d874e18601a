#!/bin/bash
# Full-length chains through the CLI with synthetic weights (no checkpoints offline): completes, finite, memory stable.
# tools/full_chain.sh <config relative to scripts/configs> [extra flags]
cd "$GRAFT_REPO_ROOT" || exit 1
cfg=$1; shift
mkdir -p /tmp/fc && cd /tmp/fc
time python "$GRAFT_REPO_ROOT/rule-guided-music_amd/scripts/sample_rule.py" --config_path "$GRAFT_REPO_ROOT/rule-guided-music_amd/scripts/configs/$cfg" \
  --model DiTRotary_XL_8 --image_size 128 16 --in_channels 4 --scale_factor 1.2465 --class_cond True --num_classes 3 --class_label 1 \
  --synthetic_weights True --progress False "$@" > /tmp/fc/log.txt 2>&1; tail -25 /tmp/fc/log.txt | cut -c1-300
python - <<'PY'
import glob, pandas as pd, numpy as np
for f in glob.glob("/tmp/fc/loggings/**/results.csv", recursive=True):
    df = pd.read_csv(f)
    print(f, len(df), {c: float(df[c].mean()) for c in df.columns if c.endswith(".loss")})
for f in glob.glob("/tmp/fc/loggings/**/sample_0*.npy", recursive=True)[:1]:
    a = np.load(f); print(f, a.shape, a.dtype, int(a.max()), float((a > 0).mean()))
PY
