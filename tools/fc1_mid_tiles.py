"""fc1 (GELU + split-row output, cold weights) at B = 6 / 8 / 12: the heuristic against explicit tiles, incl. the two-shape launches 48 / 49."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
os.environ["SWEEP_ACT"] = "2"; os.environ["SWEEP_SPLIT"] = "1"; os.environ["SWEEP_COLD"] = "1"
import gemm_sweep as g
for b in (6, 8, 12):
    row = []
    for t in (100, 144, 143, 148, 149, 171, 173):
        try:
            tf, us = g.bench(256 * b, 4608, 1152, t, iters=30, check=False)
            row.append(f"{t}: {us:6.1f}")
        except Exception as e:
            row.append(f"{t}:   n/a")
    print(f"fc1 B={b}  us  " + "  ".join(row), flush=True)
