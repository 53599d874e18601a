"""-m gpu: round-6 additions -- the chord analyser beside the GPU (reference guided_diffusion/gaussian_diffusion.py:1363-1375
blocks the step on pool.map) and the exact-fp32 final decode (reference guided_diffusion/midi_util.py:42-64)."""
import os
import time
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_sampler import SM, _diffusion, _dit, _model_fn, _vae

pytestmark = pytest.mark.gpu
F32 = np.float32


def _scg_step(rules, weights, n=16, seg=False):
    """one SCG p_sample step of the small eps-network + the real decoder on the 'steps' golden's inputs (B = 2); -> closure(d) -> results"""
    from gpu_util import dev
    from guided_diffusion.gaussian_diffusion import PhiloxNoise
    g = load_golden("steps")
    m, vae = _dit(SM, 11), _vae(2)
    guid = SimpleNamespace(schedule=True, t_start=750, t_end=0, interval=1, method="no_guidance")
    scg = dict(num_samples=n, **weights)

    def run():
        d = _diffusion("")
        d.t_end = 0
        d.noise = PhiloxNoise(seed=99)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = d.p_sample(_model_fn(m), dev(g["x"]), dev(g["scg.t"]), clip_denoised=False, model_kwargs={"y": dev(g["y"]), "rule": rules},
                         embed_model=vae, scale_factor=1.2465, guidance_kwargs=guid, scg_kwargs=scg)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out["sample"].clone(), d.last_scg["total_log_prob"].clone(), d.last_scg["max_ind"].clone()
    return run


def test_chord_analyser_runs_beside_the_gpu(monkeypatch):
    """f3: a search step with a chord rule -- the candidates decode in chunks, every chunk's uint8 rolls go to the host analyser while the
    next chunk decodes (music_rules.get_chords_async), the answers are joined in the rule loop.  Same table, winners and sample as the
    reference's blocking order; with an analyser as slow as the GPU's share of the step, the step takes clearly less than their sum."""
    import gpu_util
    from gpu_util import dev
    import guided_diffusion.gaussian_diffusion as gd
    from music_rule_guidance import music_rules
    g = load_golden("steps")
    B, n = 2, 16
    rules = {"pitch_hist": dev(g["scg.target.pitch_hist"]), "note_density": dev(g["scg.target.note_density"]),
             "chord_progression": torch.tensor([[1, 3, 5, 0, 2, 4, 6, 1]] * B, dtype=torch.long, device="cuda")}
    run = _scg_step(rules, {"pitch_hist": 40., "note_density": 1., "chord_progression": 2.})
    no_chord = _scg_step({k: v for k, v in rules.items() if "chord" not in k}, {"pitch_hist": 40., "note_density": 1.})
    try:
        music_rules.register_chord_backend(gpu_util.slow_chord_backend, workers=0)
        monkeypatch.setattr(gpu_util, "SLOW_CHORD_SLEEP", 0.0)
        for _ in range(2):
            no_chord()
        t_dev = min(no_chord()[0] for _ in range(3))                     # the GPU's share: decode of 32 candidates + the device rules
        sleep = max(t_dev, 0.02) / (n * B)
        monkeypatch.setattr(gpu_util, "SLOW_CHORD_SLEEP", sleep)            # host share == device share
        monkeypatch.setattr(gd, "CHORD_ASYNC", False)
        assert gd._async_chord_rules(rules) == []
        run()
        sync = [run() for _ in range(3)]
        monkeypatch.setattr(gd, "CHORD_ASYNC", True)
        assert gd._async_chord_rules(rules) == ["chord_progression"]
        run()
        asyn = [run() for _ in range(3)]
    finally:
        music_rules.register_chord_backend(None)
    t_sync, t_async = min(r[0] for r in sync), min(r[0] for r in asyn)
    print(f"[chord overlap] device share {1e3 * t_dev:.1f} ms, analyser {1e3 * sleep * n * B:.1f} ms: blocking {1e3 * t_sync:.1f} ms, beside the GPU {1e3 * t_async:.1f} ms")
    for a, b in zip(sync[0][1:], asyn[0][1:]):
        assert torch.equal(a, b)                                          # sample, (n, B) table, winners: identical
    assert t_sync > 1.7 * t_dev                                           # the blocking order pays device + host
    assert t_async < 0.8 * t_sync, (t_dev, t_sync, t_async)                # beside the GPU: ~ one chunk's decode + the analyser
    # the order rule: a user's rule in FRONT of the chord rule may write anything into the roll -> the reference's blocking order is kept
    from music_rule_guidance.rule_maps import FUNC_DICT
    try:
        music_rules.register_chord_backend(gpu_util.fake_chord_backend, workers=0)
        FUNC_DICT["my_rule"] = lambda roll: roll[:, 0].mean(dim=(1, 2)).unsqueeze(-1)
        assert gd._async_chord_rules({"my_rule": None, "chord_progression": None}) == []
        assert gd._async_chord_rules({"chord_progression": None, "my_rule": None}) == ["chord_progression"]
        assert gd._async_chord_rules({"chord_progression_pixel": None}) == ["chord_progression_pixel"]
    finally:
        FUNC_DICT.pop("my_rule", None)
        music_rules.register_chord_backend(None)


def test_chord_futures_return_what_get_chords_returns():
    from conftest import chord_test_roll
    from gpu_util import dev, fake_chord_backend
    from music_rule_guidance import music_rules
    g = load_golden("chord_quantise")
    try:
        for workers in (0, 2):
            music_rules.register_chord_backend(fake_chord_backend, workers=workers)
            roll = dev(chord_test_roll(int(g["seed"])))
            futs = [music_rules.get_chords_async(roll.clone(), return_key=True), music_rules.get_chords_async(roll[:1].clone())]
            want = music_rules.get_chords(roll.clone(), return_key=True)
            got = futs[0].result()
            assert torch.equal(got[0], want[0]) and got[1] == want[1] and got[2] == want[2]
            assert futs[1].result().shape == (2,)                        # N == 1 squeezes, like get_chords
            # the preamble's writes land in the roll it was given, like get_chords'
            a, b = dev(chord_test_roll(int(g["seed"]))), dev(chord_test_roll(int(g["seed"])))
            music_rules.get_chords_async(a).result()
            music_rules.get_chords(b)
            assert torch.equal(a, b)
    finally:
        music_rules.register_chord_backend(None)


@pytest.mark.parametrize("N,D,L", [(16, 1152, 195840), (1, 1152, 195840), (32, 1152, 6 * 1152), (5, 384, 27648), (17, 384, 27648), (2, 768, 6 * 768 * 2), (68, 1152, 14 * 1152)])
def test_adaln_stream_kernel_is_the_exact_fp32_product(N, D, L):
    """csrc/adaln_stream.hip: mod = SiLU(c) . W^T + b of all blocks in one weight-streaming pass (ref guided_diffusion/dit.py:333, :374)
    against the float64 product -- fp32 products, fp32 accumulation: error at the level of an fp32 dot product of D terms."""
    from rgm import native as R
    g = torch.Generator(device="cuda").manual_seed(N * 7 + D)
    cs = torch.randn(N, D, device="cuda", generator=g)
    W = torch.randn(L, D, device="cuda", generator=g) * 0.05
    b = torch.randn(L, device="cuda", generator=g)
    out = torch.full((N, L), float("nan"), device="cuda")
    R.check(R.lib.rgm_adaln_stream(R.ptr(cs), R.ptr(W), R.ptr(b), R.ptr(out), N, D, L, R.current_stream()))
    ref = cs.double() @ W.double().t() + b.double()
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err
    out2 = torch.empty_like(out)
    R.check(R.lib.rgm_adaln_stream(R.ptr(cs), R.ptr(W), None, R.ptr(out2), N, D, L, R.current_stream()))
    assert torch.equal(out2 + b, out) or (out2 + b - out).abs().max().item() < 1e-6
    # a row is the same fixed-order sum in any batch
    k = min(N, 3)
    R.check(R.lib.rgm_adaln_stream(R.ptr(cs[N - k:].contiguous()), R.ptr(W), R.ptr(b), R.ptr(out2), k, D, L, R.current_stream()))
    assert torch.equal(out2[:k], out[N - k:])
    with pytest.raises(R.RgmError):
        R.check(R.lib.rgm_adaln_stream(R.ptr(cs), R.ptr(W), R.ptr(b), R.ptr(out), 257, D, L, R.current_stream()))


@pytest.mark.parametrize("M,N,K,wide", [(4096, 4608, 1152, 74), (3584, 4608, 1152, 74), (512, 576, 64, 74), (768, 584, 96, 74)])
def test_256x288_tiles_equal_the_256x256_kernel_bit_for_bit(M, N, K, wide):
    """csrc/gemm2_body.h SBLO (round 6): the pre-split GEMM on 256 x 288 workgroup tiles -- four waves of 64 x 288, 288 accumulator registers per lane
    (256 AGPRs + 32 VGPRs pinned by asm constraints), only the hi fragment halves double-buffered -- fc1 of DiT-XL at M = 4096 (ref
    guided_diffusion/dit.py:326, 336) as ONE round of 256 tiles.  Same K order and term order (al*bh, ah*bl, ah*bh) per accumulator as the 256 x 256
    kernel: bit-identical outputs, fp32 rows and GELU + split rows; the heuristic picks it for the C2 shape."""
    import ctypes as C
    from gpu_util import dev, rel, split_torch_dtype
    from rgm import native as R
    from test_gpu_fullsize import _ref, _split
    rng = np.random.RandomState(M + N + K)
    A, B = rng.randn(M, K).astype(F32), (rng.randn(N, K) * 0.03).astype(F32)
    bias = rng.randn(N).astype(F32)
    As, Bs, bd = _split(A), _split(B), dev(bias)
    need = max(int(R.lib.rgm_gemm_scratch_bytes(M, N)), 4096)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    st = R.current_stream()
    R.set_gemm_precision("bf16x3_presplit")
    try:
        outs = {}
        for tile in (wide, 71):
            c = torch.full((M, N), float("nan"), device="cuda")
            R.check(R.lib.rgm_gemm_split_epi(R.ptr(As), K, R.ptr(Bs), K, R.ptr(c), N, M, N, K, R.ptr(bd), 0, 1.0, None, 0, 1, None, 0, tile, 0,
                                             R.ptr(ws), need, st))
            h = torch.full((M, N), float("nan"), device="cuda")
            if N % 32 == 0:                       # split rows need whole 32-column blocks
                R.check(R.lib.rgm_gemm_split_epi(R.ptr(As), K, R.ptr(Bs), K, R.ptr(h), N, M, N, K, R.ptr(bd), 2, 1.0, None, 0, 1, None, 0, tile, 1,
                                                 R.ptr(ws), need, st))
            torch.cuda.synchronize()
            outs[tile] = (c, h)
        assert torch.equal(outs[wide][0], outs[71][0])
        assert rel(outs[wide][0].cpu().numpy(), _ref(A, B, bias, 0, 1.0, None, 1, None)) < 3e-5
        if N % 32 == 0:
            assert torch.equal(outs[wide][1].view(torch.int32), outs[71][1].view(torch.int32))
            raw = outs[wide][1].view(split_torch_dtype()).view(M, N // 32, 2, 32).float().cpu().numpy().astype(np.float64)
            assert rel((raw[:, :, 0] + raw[:, :, 1]).reshape(M, N), _ref(A, B, bias, 2, 1.0, None, 1, None)) < 3e-5
        if (M, N, K) == (4096, 4608, 1152):   # the heuristic's choice for fc1 at B = 16: one launch of kernel id 40 + tile
            R.check(R.lib.rgm_prof_reset())
            R.check(R.lib.rgm_prof_enable(1))
            h2 = torch.empty((M, N), device="cuda")
            R.check(R.lib.rgm_gemm_split_epi(R.ptr(As), K, R.ptr(Bs), K, R.ptr(h2), N, M, N, K, R.ptr(bd), 2, 1.0, None, 0, 1, None, 0, 0, 1,
                                             R.ptr(ws), need, st))
            torch.cuda.synchronize()
            R.check(R.lib.rgm_prof_enable(0))
            n, ms, fl = C.c_int(0), C.c_double(0), C.c_double(0)
            R.check(R.lib.rgm_prof_report(40 + wide, C.byref(n), C.byref(ms), C.byref(fl)))
            R.check(R.lib.rgm_prof_reset())
            assert n.value == 1, n.value
            assert torch.equal(h2.view(torch.int32), outs[wide][1].view(torch.int32))
    finally:
        R.set_gemm_precision("fp32")


@pytest.mark.parametrize("workers", [0, 2])
def test_cli_with_a_chord_analyser_runs_the_reference_config_whole(tmp_path, monkeypatch, workers):
    """scripts/sample_rule.py on cond_table/all/scg_classifier_all.yml AS SHIPPED BY THE REFERENCE -- three classifiers (pitch, note density, chords) AND
    SCG over three rules incl. chord_progression -- with an analyser registered through --chord_backend (a stand-in with the reference's
    piano_roll_to_chords signature; music21 is not installable here): the chord rule is scored beside the GPU in every search step (in this process and in
    the spawn-context worker pool), nothing is dropped, the report carries the chord columns."""
    import json
    import pandas as pd
    from music_rule_guidance import music_rules
    from test_gpu_cli import CFG, COMMON, _cli
    monkeypatch.chdir(tmp_path)
    cli = _cli()
    cfg = os.path.join(CFG, "cond_table/all/scg_classifier_all.yml")
    try:
        res = cli.main(["--config_path", cfg, "--batch_size", "2", "--num_samples", "2", "--diffusion_steps", "25", "--chord_backend",
                        "gpu_util:fake_chord_backend", "--chord_workers", str(workers)] + COMMON)
    finally:
        music_rules.register_chord_backend(None)
    out_dir = os.path.join("loggings", cli.output_dir_for(cfg, 1))
    meta = json.load(open(os.path.join(out_dir, "run_metadata.json")))
    assert not meta["dropped_rules"]
    df = pd.read_csv(os.path.join(out_dir, "results.csv"))
    assert len(df) == 2 and len(res) == 2
    for r in ("pitch_hist", "note_density", "chord_progression"):
        assert {f"{r}.target_rule", f"{r}.gen_rule", f"{r}.loss"} <= set(df.columns), df.columns
        assert np.isfinite(df[f"{r}.loss"]).all()


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16x3_presplit"])
def test_conditioning_computed_ahead_is_the_same_forward(precision, monkeypatch):
    """guided_diffusion/dit.py cond_hint / rgm_dit_cond_rows / rgm_dit_forward_cond: the adaLN modulation of a sample is a function of (t, y)
    (ref dit.py:621-628, :333, :374), and a loop knows the timesteps it will visit -- the rows of the next steps, for every label of the table,
    come from ONE pass over the adaLN weights and the forwards gather theirs.  Must be the same numbers, bit for bit: (a) a single forward with
    and without the hint, (b) a whole DDIM-8 chain with labels and a DDPM chain without, RGM_COND_AHEAD on and off, (c) the weight pass really
    runs once per group of steps, and parameters loaded later invalidate the rows."""
    from gpu_util import dev
    from guided_diffusion import dit as dit_mod
    from guided_diffusion.gaussian_diffusion import PhiloxNoise
    from rgm import native as R
    R.set_gemm_precision(precision)
    try:
        m = _dit(SM, 11, final_std=0.05)
        rng = np.random.RandomState(3)
        x = dev(rng.randn(5, 4, 128, 16).astype(F32))
        y = dev(np.array([0, 2, 1, 3, 0], dtype=np.int64))
        t = torch.full((5,), 620, dtype=torch.int64, device="cuda")
        monkeypatch.setattr(dit_mod, "COND_AHEAD", 32)
        plain = m(x, t, y).clone()
        with dit_mod.cond_hint((620, [620, 600, 580])):
            ahead = m(x, t, y).clone()
            ahead_nolabel = m(x, t, None).clone()
        assert torch.equal(plain, ahead)
        assert torch.equal(m(x, t, None), ahead_nolabel)
        # rows of explicit (t, y) pairs == the rows a forward uses: sample 1 above is (620, label 2)
        rows = m.cond_rows([620, 600], [2, 2], 128)
        assert rows.shape == (2, (6 * SM["depth"] + 2) * SM["hidden"]) and bool(torch.isfinite(rows).all())

        calls = []
        real = dit_mod.DiTRotary.cond_rows
        monkeypatch.setattr(dit_mod.DiTRotary, "cond_rows", lambda self, ts, ys, H: (calls.append(len(ts)), real(self, ts, ys, H))[1])

        def chain(on, labels):
            monkeypatch.setattr(dit_mod, "COND_AHEAD", 32 if on else 0)
            d = _diffusion("ddim8" if labels else "8")
            d.noise = PhiloxNoise(seed=5)
            kw = {"y": y[:3]} if labels else {}
            fn = _model_fn(m) if labels else (lambda xx, tt, **k: m(xx, tt, None))
            loop = d.ddim_sample_loop if labels else d.p_sample_loop
            return loop(fn, (3, 4, 128, 16), noise=x[:3].clone(), clip_denoised=False, model_kwargs=kw, device="cuda", progress=False).clone()

        for labels in (True, False):
            calls.clear()
            a = chain(True, labels)
            n_on = list(calls)
            b = chain(False, labels)
            assert torch.equal(a, b), (labels, float((a - b).abs().max()))
            # 8 steps: with labels 4 rows per step -> 8 steps fit one pass of 32 rows; without labels 1 row per step
            assert n_on == ([32] if labels else [8]), n_on
            assert len(calls) == len(n_on)                      # (the chain with the switch off never asks for rows)
        # new parameters: the rows computed ahead belong to the old ones
        monkeypatch.setattr(dit_mod, "COND_AHEAD", 32)
        m2 = _dit(SM, 12, final_std=0.05)
        with dit_mod.cond_hint((620, [620, 600])):
            before = m(x, t, y).clone()
            m.load_state_dict(m2.state_dict())
            after = m(x, t, y).clone()
        assert torch.equal(after, m2(x, t, y)) and not torch.equal(before, after)
    finally:
        R.set_gemm_precision("fp32")
