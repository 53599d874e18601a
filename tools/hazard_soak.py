"""Soak for the instruction pattern tools/isa_lint.py reports (a multi-dword VMEM load whose last destination register is read right behind
its s_waitcnt vmcnt(0): DESIGN 4h): the kernels that carry it AND share their CUs with lock-step twins of themselves -- LayerNorm,
K-slice reduce, the 128-row GEMMs, the classifier chain, GroupNorm passes, the sampler's element-wise kernels -- are run N times on the same
inputs; every run must equal the first bit for bit.  The attention instance of the hazard showed in ~1 workgroup of 4000; one XL-28 forward
at B = 16 alone launches ~10^5 workgroups of these kernels.
    python tools/hazard_soak.py [N]        (default 100; tests/test_gpu_round5.py runs a short one)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rule-guided-music_amd"), os.path.join(ROOT, "tests")]


def soak(N, log=print):
    import numpy as np
    import torch
    from rgm import native as R, synth
    from gpu_util import load_module
    from guided_diffusion.dit import DiTRotary, DiTRotaryClassifier
    from guided_diffusion.gaussian_diffusion import _decode
    from taming.models.klvae_pedal import AutoencoderKL
    F32 = np.float32
    rng = np.random.RandomState(11)
    arch = dict(depth=28, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4, num_classes=3)
    m = load_module(DiTRotary(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=1152, depth=28, num_heads=16, num_classes=3,
                              learn_sigma=False), synth.dit_state_dict(1, final_std=0.3 / 1152 ** 0.5, device="cuda", **arch))
    carch = dict(depth=12, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=16)
    clf = load_module(DiTRotaryClassifier(input_size=[128, 16], patch_size=8, in_channels=4, hidden_size=384, depth=12, num_heads=6,
                                          num_classes=16), synth.dit_state_dict(3, **carch))
    vae = load_module(AutoencoderKL(), synth.vae_state_dict(2, encoder=True))

    def inputs(B):
        return (torch.from_numpy(rng.randn(B, 4, 128, 16).astype(F32)).cuda(), torch.from_numpy(rng.randint(0, 1000, size=B).astype(np.int64)).cuda(),
                torch.from_numpy(rng.randint(0, 3, size=B).astype(np.int64)).cuda())
    x16, x4, x5 = inputs(16), inputs(4), inputs(5)
    tgt = torch.from_numpy(rng.rand(4, 16).astype(F32) * 4).cuda()
    z = torch.from_numpy(rng.randn(8, 4, 128, 16).astype(F32) * 0.8).cuda()
    cases = [
        ("XL-28 forward B=16, pre-split (256x256 / 128x64 tiles, ln_mod, splitk_reduce_ln)", "bf16x3_presplit", lambda: m(*x16)),
        ("XL-28 forward B=4, pre-split (small-grid tiles, gemm144, splitk_reduce)", "bf16x3_presplit", lambda: m(*x4)),
        ("XL-28 forward B=5, on-the-fly bf16x3 (gemm_kernel)", "bf16x3", lambda: m(*x5)),
        ("XL-28 forward B=4, fp32 MFMA (gemm_kernel PREC 0, fp32 attention)", "fp32", lambda: m(*x4)),
        ("classifier value-and-grad B=4 (ln_mod_bwd, attention backward, gate_rows, loss_grad)", "bf16x3_presplit",
         lambda: torch.cat([t_.reshape(-1) for t_ in clf.value_and_grad(x4[0], x4[1], tgt, "mse", 10.0)])),
        ("KL-VAE decode of 8 latents (GroupNorm passes, conv_in / conv_out, implicit-conv tiles)", "bf16x3_presplit", lambda: _decode(z, vae, 1.0)),
    ]
    bad = 0
    for name, prec, fn in cases:
        R.set_gemm_precision(prec)
        first = fn().clone()
        torch.cuda.synchronize()
        diff = 0
        for k in range(N):
            if not torch.equal(fn(), first):
                diff += 1
        torch.cuda.synchronize()
        log(f"{name}: {N} runs, {diff} differ from the first")
        bad += diff
    R.set_gemm_precision("fp32")
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    sys.exit(1 if soak(n) else 0)
