import os, sys
sys.path.insert(0, "/root/repo/tools")
os.environ["SWEEP_ACT"] = "2"; os.environ["SWEEP_SPLIT"] = "1"; os.environ["SWEEP_COLD"] = "1"
import gemm_sweep as g
tiles = [143, 144, 146, 152, 153, 154, 156, 157, 158, 105]
print("fc1 remainder (4096,512,1152) GELU+split, cold weights")
for t in tiles:
    try:
        tf, us = g.bench(4096, 512, 1152, t, iters=30, check=False)
        print(f"tile {t}: {us:7.1f} us {tf:6.1f} TF", flush=True)
    except Exception as e:
        print("tile", t, "failed", str(e)[:80])
os.environ["SWEEP_ACT"] = "0"; os.environ["SWEEP_SPLIT"] = "0"
import importlib; importlib.reload(g)
print("proj (4096,1152,1152) plain")
for t in tiles:
    try:
        tf, us = g.bench(4096, 1152, 1152, t, iters=30, check=False)
        print(f"tile {t}: {us:7.1f} us {tf:6.1f} TF", flush=True)
    except Exception as e:
        print("tile", t, "failed", str(e)[:80])
