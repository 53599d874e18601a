import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/rule-guided-music_amd")
from rgm import native as R
st = R.current_stream()
g = torch.Generator(device="cuda").manual_seed(0)
K = 1152
for Nn in (6912, 195840):
    b = torch.randn(Nn, K, device="cuda", generator=g) * 0.03
    bias = torch.randn(Nn, device="cuda", generator=g)
    for M in (8, 16, 17, 24, 32, 48, 64, 100):
        a = torch.randn(M, K, device="cuda", generator=g)
        for prec in ("bf16x3", "fp32"):
            R.set_gemm_precision(prec)
            outs = []
            for _ in range(10):
                c = torch.empty(M, Nn, device="cuda")
                R.check(R.lib.rgm_gemm(R.ptr(a), K, R.ptr(b), K, R.ptr(c), Nn, M, Nn, K, R.ptr(bias), 0, 1.0, None, 0, 1, None, 0, st))
                outs.append(c)
            torch.cuda.synchronize()
            sp = max(float((o - outs[0]).abs().max()) for o in outs)
            ref = (a.double() @ b.double().t() + bias.double())
            err = float((outs[0].double() - ref).abs().max() / ref.abs().max())
            rows = sorted({int(i) for o in outs for i in torch.unique(torch.nonzero((o - outs[0]).abs() > 0)[:, 0]).tolist()})
            print(f"{prec} M={M} N={Nn}: spread {sp:.2e} err {err:.2e} rows {rows[:12]}", flush=True)
