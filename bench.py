#!/usr/bin/env python3
"""bench.py -- denoising steps/s of the rule-guided sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no torchrun environment re-executes itself through torch.distributed.run with N
ranks on 127.0.0.1 (and fails loudly when the box has fewer than N devices); `n_gpus` in the JSON is the RCCL world size
that actually ran.

Metric (BASELINE.json): denoising steps/sec, whole node, DiTRotary_XL_8 on 4x128x16 latents.
Headline workload = BASELINE config[1] ("C2"): unconditional DDIM-50 (eta=1) on a batch of 16 latents per GPU, one "step" =
one ddim_sample call over that batch (eps-network forward + fused step update + Philox noise).  W untimed warm-up steps,
then three timed regions of exactly K steps each, each bracketed by barrier + device synchronise, MAX over ranks per
region; `value` / `ms_per_step` are the MEDIAN region (all three are listed in config.repeats_ms_per_step).  For N > 1 every
rank runs its own batch-16 chain (the path shards over samples; no data-path collective): scaling = weak.
Nothing of a step is skipped or carried over: the one thing the sampler computes for several steps at once -- the adaLN modulation rows of
the schedule's next timesteps, one pass over those weights per 32 rows (guided_diffusion/dit.py cond_hint; bit-identical chains) -- is
dropped at the start of every timed region, so each region pays the passes its K steps need (config.cond_ahead counts them;
RGM_COND_AHEAD=0 runs the pass in every forward).

One JSON line on stdout (rank 0) per the driver contract, plus
  roofline     : the dominant kernel of the headline run (pre-split bf16x3 GEMM) -- algorithmic 2MNK FLOPs of its launches
                 divided by their HIP-event durations measured live on the launch stream (separate 2-step pass with
                 rgm_prof_enable, same workload), against 2500 / 3 TFLOP/s (dense bf16 MFMA peak, 3 MFMAs per product);
  fp32_exact   : the same C2 step in exact-fp32 MFMA arithmetic (v_mfma_f32_32x32x2_f32): steps/s, ms/step and its dominant
                 kernel against the 157.3 TFLOP/s f32 matrix peak;
  scg          : the north star's sharded step -- BASELINE config[3] shape, B = 4, n = 16 candidates, DiT-XL + KL-VAE decode +
                 2 rules -- STRONG scaling over the N ranks (candidates partitioned, one RCCL all-gather of the (n, B)
                 log-probabilities per step): ms/step, the all-gather's share, and whether every rank picked the same winners;
  config.uint8_flips : the 50-step B = 2 XL-28 golden chain of tests/golden (reference-generated): entries of the decoded
                 uint8 roll that differ from the reference's, and how many of them are NOT on a quantisation boundary;
  cpu_baseline : BASELINE config[0] ("C1": the same DDIM-50 chain at batch 2) on this box's host cores with the torch-CPU
                 restatement oracle/dit_torch.py (kind "port": the reference itself cannot travel), median of 3 steps.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "rule-guided-music_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

XL = dict(depth=28, hidden=1152, heads=16, patch=8, in_ch=4, out_ch=4)
DIT_GFLOP_PER_SAMPLE = 237.4          # DiTRotary_XL_8 forward, T=256
DIT_GFLOP_PER_HALF_WINDOW = 116.8     # the same forward at T=128 (DiffCollage half window)
VAE_GFLOP_PER_TILE = 114.48
CLS_VAG_GFLOP_PER_SAMPLE = 25.4       # DiTRotary-S/8-cls forward + dgrad
F32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH: v_mfma_f32_32x32x2_f32, dense
BF16X3_EQUIV_PEAK_TFLOPS = 2500.0 / 3  # dense bf16 MFMA peak, 3 MFMAs per useful (algorithmic) product


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_eps_model(num_classes, device):
    from rgm import synth
    from guided_diffusion.dit import DiT_models
    m = DiT_models["DiTRotary_XL_8"](input_size=[128, 16], in_channels=4, num_classes=num_classes, learn_sigma=False)
    arch = dict(XL, num_classes=num_classes + (1 if num_classes else 0), class_dropout=False)
    sd = synth.dit_state_dict(1, final_std=0.3 / 1152 ** 0.5, device=device, **arch)
    m.load_state_dict(sd, strict=True)
    return m.to(device).eval()


def make_diffusion(respacing):
    from guided_diffusion.script_util import create_diffusion
    return create_diffusion(learn_sigma=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing=respacing,
                            use_kl=False, predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False)


class C2Workload:
    """Unconditional DDIM-50, eta = 1, batch 16 (sample_rule.py with num_classes=0, class_cond False)."""
    name = "C2 unconditional DDIM-50 (eta=1), DiTRotary_XL_8, batch 16 per GPU, latent 4x128x16"

    def __init__(self, device, batch):
        from functools import partial
        from guided_diffusion.condition_functions import model_fn
        self.B = batch
        self.device = device
        self.model = build_eps_model(0, device)
        self.fn = partial(model_fn, model=self.model, num_classes=0, class_cond=False, cfg=False, w=0.)
        self.d = make_diffusion("ddim50")
        self.d.t_end = 0
        self.x = self.d._draw((batch, 4, 128, 16), device)
        self.k = 0
        self.flop_per_step = batch * DIT_GFLOP_PER_SAMPLE * 1e9

    def step(self):
        i = 49 - (self.k % 50)
        self.k += 1
        if i == 49:
            self.x = self.d._draw(self.x.shape, self.device)      # new chain
        t = torch.full((self.B,), i, dtype=torch.int64, device=self.device)
        self.d._t_host = i
        out = self.d.ddim_sample(self.fn, self.x, t, clip_denoised=False, model_kwargs={}, eta=1.0)
        self.d._t_host = None
        self.x = out["sample"]


class SCGWorkload:
    """One guided step of BASELINE config[3] = cond_table/all/scg_classifier_all.yml as the reference defines it (:6-35), minus the
    music21 chord rule: classifier guidance with the pitch-histogram and note-density DiTRotary-S/8-cls (scales 400 / 10) AND SCG with
    n = 16 candidates over B = 4 -- DiT-XL on 4 + 64 rows, KL-VAE decode of 512 squares, 2 rules, DDPM chain.  SURVEY 8d prices it
    (1 + 16) * 237.4 + 16 * 915.8 + 2 * 25.4 GFLOP per sample."""
    name = ("C4 classifier-guided (pitch_hist + note_density DiTRotary-S/8-cls) + SCG DDPM step, DiTRotary_XL_8 + KL-VAE decode + "
            "pitch_hist/note_density rules, B=4, n=16 (sharded over GPUs)")
    classifiers = True

    def __init__(self, device, batch):
        from functools import partial
        from types import SimpleNamespace
        from rgm import synth
        from guided_diffusion.dit import DiT_models
        from guided_diffusion.condition_functions import model_fn, composite_nn_zt
        from taming.models.klvae_pedal import AutoencoderKL
        self.B = batch
        self.device = device
        self.model = build_eps_model(3, device)
        self.fn = partial(model_fn, model=self.model, num_classes=3, class_cond=True, cfg=False, w=0.)
        self.vae = AutoencoderKL()
        self.vae.load_state_dict(synth.vae_state_dict(2, device=device), strict=False)   # decoder slice; the encoder keeps its init
        self.vae = self.vae.to(device).eval()
        self.d = make_diffusion("")
        self.d.t_end = 0
        self.d.noise = __import__("guided_diffusion.gaussian_diffusion", fromlist=["PhiloxNoise"]).PhiloxNoise(seed=0)
        self.x = self.d._draw((batch, 4, 128, 16), device)
        ph = torch.tensor([0.5, 0, 0, 0, 0.25, 0, 0, 0.25, 0, 0, 0, 0], device=device).repeat(batch, 1)
        nd = torch.tensor([3.] * 8 + [3.] * 8, device=device).repeat(batch, 1)
        self.kw = {"y": torch.ones(batch, dtype=torch.int64, device=device), "rule": {"pitch_hist": ph, "note_density": nd}}
        self.cond = None
        cls_gflop = 0.0
        if self.classifiers:
            self.clfs = []
            for k, seed in ((12, 5), (16, 3)):
                clf = DiT_models["DiTRotary-S/8-cls"](input_size=[128, 16], in_channels=4, num_classes=k)
                arch = dict(depth=12, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=k)
                clf.load_state_dict(synth.dit_state_dict(seed, device=device, **arch))
                self.clfs.append(clf.to(device).eval())
            self.cond = partial(composite_nn_zt, fns=["grad_nn_zt_mse", "grad_nn_zt_mse"], classifier_scales=[400., 10.],
                                classifiers=self.clfs, rule_names=["pitch_hist", "note_density"])
            cls_gflop = 2 * CLS_VAG_GFLOP_PER_SAMPLE
        self.guid = SimpleNamespace(schedule=True, t_start=750, t_end=0, interval=1,
                                    method="classifier_guidance" if self.classifiers else "no_guidance")
        self.scg = {"num_samples": 16, "pitch_hist": 40., "note_density": 1.}
        self.k = 0
        self.flop_per_step = batch * ((1 + 16) * DIT_GFLOP_PER_SAMPLE + 16 * 8 * VAE_GFLOP_PER_TILE + cls_gflop) * 1e9

    def step(self):
        i = 700 - (self.k % 600)
        self.k += 1
        t = torch.full((self.B,), i, dtype=torch.int64, device=self.device)
        self.d._t_host = i
        out = self.d.p_sample(self.fn, self.x, t, clip_denoised=False, cond_fn=self.cond, model_kwargs=self.kw, embed_model=self.vae,
                              scale_factor=1.2465, guidance_kwargs=self.guid, scg_kwargs=self.scg)
        self.d._t_host = None
        self.x = out["sample"]


class LongWorkload(SCGWorkload):
    """BASELINE config[4]: a 40.96 s sequence -- linear DiffCollage over 7 windows of 128 latent rows with overlap 64
    (latent 4 x 512 x 16) -- with SCG (n = 16) scoring rule(decode(x0)) on the 4096-frame roll."""
    name = ("C5 DiffCollage (linear, 7 windows, overlap 64) + SCG guided DDPM step, DiTRotary_XL_8 + KL-VAE decode of 32 squares "
            "per candidate + pitch_hist/note_density, n=16 (sharded over GPUs)")
    classifiers = False     # BASELINE config[4] is collage + SCG; the 128-row classifiers do not apply to a 512-row latent

    def __init__(self, device, batch):
        from functools import partial
        import diff_collage as dc
        from guided_diffusion.condition_functions import dc_model_fn
        super().__init__(device, batch)
        model = self.model

        def eps_fn(x, t, y=None):            # the backbone takes (4, time, pitch)
            return model(x.permute(0, 1, 3, 2).contiguous(), t, y=y).permute(0, 1, 3, 2)
        worker = dc.CondIndSimple((4, 16, 128), eps_fn, 7, overlap_size=64)
        self.fn = partial(dc_model_fn, model=worker.eps_scalar_t_fn, num_classes=3, class_cond=True, cfg=False, w=0.)
        H = worker.shape[2]
        self.x = self.d._draw((batch, 4, H, 16), device)
        nw = H * 8 // 128
        ph = torch.tensor([0.5, 0, 0, 0, 0.25, 0, 0, 0.25, 0, 0, 0, 0], device=device).repeat(batch, 1)
        nd = torch.tensor([3.] * nw + [3.] * nw, device=device).repeat(batch, 1)
        self.kw = {"y": torch.ones(batch, dtype=torch.int64, device=device), "rule": {"pitch_hist": ph, "note_density": nd}}
        per_forward = 7 * DIT_GFLOP_PER_SAMPLE + 6 * DIT_GFLOP_PER_HALF_WINDOW   # 7 full windows + the 6 overlap halves (T = 128) of the collage
        self.flop_per_step = batch * ((1 + 16) * per_forward + 16 * (H // 16) * VAE_GFLOP_PER_TILE) * 1e9


class C3Workload:
    """Classifier guidance only (config[2]): p_sample on the '250' chain, batch 32, one DiTRotary-S/8-cls (note density)."""
    name = "C3 classifier-guided DDPM step ('250' chain), DiTRotary_XL_8 + DiTRotary-S/8-cls value-and-grad, batch 32"

    def __init__(self, device, batch):
        from functools import partial
        from types import SimpleNamespace
        from rgm import synth
        from guided_diffusion.dit import DiT_models
        from guided_diffusion.condition_functions import model_fn, composite_nn_zt
        self.B = batch
        self.device = device
        self.model = build_eps_model(0, device)
        self.fn = partial(model_fn, model=self.model, num_classes=0, class_cond=False, cfg=False, w=0.)
        clf = DiT_models["DiTRotary-S/8-cls"](input_size=[128, 16], in_channels=4, num_classes=16)
        arch = dict(depth=12, hidden=384, heads=6, patch=8, in_ch=4, classifier=True, cls_classes=16)
        clf.load_state_dict(synth.dit_state_dict(3, device=device, **arch))
        self.clf = clf.to(device).eval()
        self.cond = partial(composite_nn_zt, fns=["grad_nn_zt_mse"], classifier_scales=[10.], classifiers=[self.clf],
                            rule_names=["note_density"])
        self.d = make_diffusion("250")
        self.d.t_end = 0
        self.x = self.d._draw((batch, 4, 128, 16), device)
        self.kw = {"rule": {"note_density": torch.tensor([3.] * 16, device=device).repeat(batch, 1)}}
        self.guid = SimpleNamespace(schedule=False, method="classifier_guidance")
        self.k = 0
        self.flop_per_step = batch * (DIT_GFLOP_PER_SAMPLE + 25.4) * 1e9

    def step(self):
        i = 249 - (self.k % 250)
        self.k += 1
        t = torch.full((self.B,), i, dtype=torch.int64, device=self.device)
        self.d._t_host = i
        out = self.d.p_sample(self.fn, self.x, t, clip_denoised=False, cond_fn=self.cond, model_kwargs=self.kw,
                              guidance_kwargs=self.guid)
        self.d._t_host = None
        self.x = out["sample"]


class DPSRuleWorkload(SCGWorkload):
    """One DPS step through the rule itself (cond_table/single/dps_rule/pitch.yml): eps-network forward with saves,
    x0 -> VAE decode with saves -> pitch_hist log p and roll gradient -> decoder input-gradient pass -> eps-network VJP."""
    classifiers = False
    name = ("dps_rule guided DDPM step ('250' chain), DiTRotary_XL_8 forward+VJP + KL-VAE decode+input-gradient + pitch_hist "
            "value-and-grad, batch 16")

    def __init__(self, device, batch):
        from functools import partial
        from types import SimpleNamespace
        from guided_diffusion.condition_functions import composite_rule
        super().__init__(device, batch)
        self.d = make_diffusion("250")
        self.d.t_end = 0
        self.cond = partial(composite_rule, fns=["rule_x0_mse_dummy"], classifier_scales=[1.], rule_names=["pitch_hist"])
        self.kw = {"y": self.kw["y"], "rule": {"pitch_hist": self.kw["rule"]["pitch_hist"]}}
        self.guid = SimpleNamespace(schedule=False, method="dps", step_size=1.0, nn=False, vae=True)
        self.flop_per_step = batch * (3 * DIT_GFLOP_PER_SAMPLE + 3 * 8 * VAE_GFLOP_PER_TILE) * 1e9

    def step(self):
        i = 249 - (self.k % 250)
        self.k += 1
        t = torch.full((self.B,), i, dtype=torch.int64, device=self.device)
        self.d._t_host = i
        out = self.d.p_sample(self.fn, self.x, t, clip_denoised=False, cond_fn=self.cond, model_kwargs=self.kw,
                              embed_model=self.vae, scale_factor=1.2465, guidance_kwargs=self.guid)
        self.d._t_host = None
        self.x = out["sample"]


def roofline_pass(work, steps=2):
    """Dominant-kernel roofline from HIP events around every GEMM launch (its own short pass)."""
    from rgm import native as R
    # per-launch durations must be exclusive: with the blocks as two half batches on two streams (rgm_set_dit_halves, B = 32 / 64 / 8 ...)
    # the launches of the halves overlap in time and every one of them reads as long as the pair -- this pass runs the single-stream forward
    prev = C.c_int(0)
    R.check(R.lib.rgm_set_dit_halves(0, C.byref(prev)))
    try:
        R.check(R.lib.rgm_prof_reset())
        R.check(R.lib.rgm_prof_enable(1))
        for _ in range(steps):
            work.step()
        torch.cuda.synchronize()
        R.check(R.lib.rgm_prof_enable(0))
    finally:
        R.check(R.lib.rgm_set_dit_halves(prev.value, None))
    rows = {}
    for kid in range(1, 140):
        n, ms, fl = C.c_int(), C.c_double(), C.c_double()
        R.check(R.lib.rgm_prof_report(kid, C.byref(n), C.byref(ms), C.byref(fl)))
        if n.value:
            rows[kid] = dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=float(R.lib.rgm_prof_bytes(kid)) if kid >= 40 else 0.0)
    R.check(R.lib.rgm_prof_reset())
    # kernel ids (csrc/gemm.hip, gemm2.hip): < 40 gemm_kernel<BM,BN,WM,WN,ALOAD,PREC> = tile + 10*ALOAD + 20*PREC;
    # >= 40 gemm2_kernel<BM,BN,WM,WN,ALOAD,NSTAGE,0,PIPE> = 40 + tile + 10*ALOAD (tiles: gemm2_launch)
    tiles = {1: "128,128,2,2", 2: "128,64,2,2", 3: "64,64,2,2", 4: "32,128,1,4", 5: "256,128,4,2", 7: "256,256,2,2", 8: "512,128,4,1", 9: "128,256,1,4", 10: "256,288,4,1"}
    g2 = {1: (1, 3, 0), 2: (2, 3, 0), 3: (3, 3, 0), 5: (5, 3, 0), 21: (1, 2, 1), 22: (2, 2, 1),
          43: (1, 2, 3), 44: (2, 2, 3), 45: (5, 2, 3), 46: (3, 3, 3), 51: (1, 3, 4), 52: (2, 3, 4),
          53: (2, 6, 4), 54: (1, 4, 4), 55: (1, 5, 4), 56: (2, 4, 4), 57: (3, 6, 4), 58: (3, 3, 4), 71: (7, 2, 5), 72: (8, 2, 5), 73: (9, 2, 5), 74: (10, 2, 5)}

    def kname(k):
        if k == 135:
            return "gemm144_kernel<4,0,0,8>"    # csrc/gemm144.hip: 128x144 tiles on 16x16x32 MFMAs (tile 81; NSTAGE 4, L2 prefetch distance 8 -- the name rocprofv3 reports)
        if k < 40:
            return f"gemm_kernel<{tiles[k % 10]},{(k // 10) % 2},{k // 20}>"
        t = k - 40
        aload = 0
        if t not in g2 and (t - 10) in g2:
            t, aload = t - 10, 1
        shape, nstage, pipe = g2[t]
        return f"gemm2_kernel<{tiles[shape]},{aload},{nstage},0,{pipe}>"
    names = {k: kname(k) for k in rows}
    kid = max(rows, key=lambda k: rows[k]["ms"])
    r = rows[kid]
    bf16x3 = kid >= 20
    achieved = r["flops"] / (r["ms"] * 1e-3) / 1e12
    peak = BF16X3_EQUIV_PEAK_TFLOPS if bf16x3 else F32_MFMA_PEAK_TFLOPS
    traffic = None
    tp = os.path.join(ROOT, "profiles", "traffic.json")       # HBM bytes/launch from separate rocprofv3 --pmc passes
    if os.path.exists(tp):
        try:
            traffic = json.load(open(tp)).get(names[kid])
        except Exception:
            traffic = None
    all_ms = sum(v["ms"] for v in rows.values())
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic, "kernel": names[kid],
            "peak_note": ("2500 TFLOP/s dense bf16 MFMA / 3 products per fp32-equivalent FMA" if bf16x3
                          else "f32-input MFMA v_mfma_f32_32x32x2_f32"),
            "avg_launch_us": round(1e3 * r["ms"] / r["launches"], 2), "launches_per_step": r["launches"] // steps,
            "gflop_per_launch": round(r["flops"] / r["launches"] / 1e9, 3),
            "algorithmic_bytes_per_launch": round(r["bytes"] / r["launches"]) if r.get("bytes") else None,
            "share_of_gemm_time": round(r["ms"] / all_ms, 3),
            "all_gemm_tflops": round(sum(v["flops"] for v in rows.values()) / (all_ms * 1e-3) / 1e12, 2),
            # every GEMM kernel of the step with at least 2 % of the GEMM time: [name, launches per step, average us, TFLOP/s, fraction of its peak]
            "by_kernel": [[names[k], v["launches"] // steps, round(1e3 * v["ms"] / v["launches"], 1),
                           round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                           round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / (BF16X3_EQUIV_PEAK_TFLOPS if k >= 20 else F32_MFMA_PEAK_TFLOPS), 3)]
                          for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] >= 0.02 * all_ms]}


def power_pass(work, seconds=2.0):
    """Socket power and shader clock while the workload's steps run back to back for `seconds` (its own pass, after the timed regions):
    `rocm-smi --showpower --showclocks` sampled from a thread.  The one-wave-per-SIMD GEMMs sit at the board's power limit on real
    operands (profiles/r03_power_probe.txt): the clock the board sustains, not the nominal 2.4 GHz, is what the MFMA peak scales with."""
    import subprocess
    import threading
    stop, out = threading.Event(), []

    def sampler():
        while not stop.is_set():
            try:
                t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                pw = [float(l.split(":")[-1]) for l in t.splitlines() if "Power (W)" in l]
                ck = [float(l.split("(")[-1].split("Mhz")[0]) for l in t.splitlines() if "sclk" in l]
                if pw and ck:
                    out.append((pw[0], ck[0]))
            except Exception:
                return

    for _ in range(3):
        work.step()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.time()
    while time.time() - t0 < seconds:
        work.step()
        torch.cuda.synchronize()
    stop.set()
    th.join(timeout=15)
    busy = sorted(out[1:])                                      # the first sample may predate the load
    if len(busy) < 2:
        return None
    pw = sorted(p for p, _ in busy)[len(busy) // 2]
    ck = sorted(c for _, c in busy)[len(busy) // 2]
    return {"socket_power_w_median": pw, "sclk_mhz_median": ck, "samples": len(busy),
            "how": "rocm-smi --showpower --showclocks sampled while the steps of this workload ran back to back for %.1f s" % seconds}


def measure_traffic(kernel, args):
    """HBM bytes per launch of `kernel` measured IN SITU: this same workload re-run as a child under `rocprofv3 --pmc` (FETCH_SIZE and
    WRITE_SIZE in SEPARATE passes, --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), every dispatch of the
    kernel inside the sampling steps averaged -- all the shapes it runs on, weights cold as in the step -- with the guide's gfx950
    corrections (counters in KiB; FETCH_SIZE counts half of a wide coalesced read stream).  -> (bytes per launch, detail) or (None, why)."""
    import glob
    import re
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    want = re.sub(r"\s+", "", kernel)
    out = {}
    busy = None
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"):
        d = tempfile.mkdtemp(prefix="rgm_pmc_", dir="/tmp")
        cmd = [exe, "--pmc"] + counter.split() + ["--kernel-trace", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--traffic-child",
               "--workload", args.workload, "--precision", args.precision, "--steps", "3", "--warmup", "1"]
        if args.batch:
            cmd += ["--batch", str(args.batch)]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=420, check=True)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            cur = sqlite3.connect(dbs[0]).cursor()
            if " " in counter:      # matrix-pipe duty of the kernel: MFMA-busy cycles (summed over the 1024 SIMDs) over the cycles it had (summed over 8 XCDs)
                both = {}
                for cn in counter.split():
                    v = [x for n, x in cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (cn,)) if want in re.sub(r"\s+", "", n)]
                    both[cn] = float(np.mean(v)) if v else None
                shutil.rmtree(d, ignore_errors=True)
                if both.get("SQ_VALU_MFMA_BUSY_CYCLES") and both.get("GRBM_GUI_ACTIVE"):
                    busy = {"value": round(both["SQ_VALU_MFMA_BUSY_CYCLES"] / (both["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4),
                            "SQ_VALU_MFMA_BUSY_CYCLES": round(both["SQ_VALU_MFMA_BUSY_CYCLES"]), "GRBM_GUI_ACTIVE": round(both["GRBM_GUI_ACTIVE"]),
                            "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), in-situ launches averaged; 3 bf16 MFMAs per "
                                       "fp32-equivalent product, so busy * 2500 TFLOP/s * sustained/nominal clock / 3 is the kernel's fp32-equivalent rate"}
                continue
            rows = cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
        except Exception as e:
            shutil.rmtree(d, ignore_errors=True)
            if " " in counter:
                continue                     # the duty counters are an extra: never lose the traffic figure over them
            return None, f"{counter} pass failed: {e!r}"
        shutil.rmtree(d, ignore_errors=True)
        vals = [v for n, v in rows if want in re.sub(r"\s+", "", n)]
        if not vals:
            return None, f"{counter}: no dispatch of {kernel} in the profiled run"
        out[counter] = (float(np.mean(vals)), len(vals), float(np.min(vals)), float(np.max(vals)))
    fetch = 2.0 * out["FETCH_SIZE"][0] * 1024.0
    write = out["WRITE_SIZE"][0] * 1024.0
    detail = {"method": "rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes) over this workload, all in-situ launches of the kernel",
              "fetch_bytes_corrected": round(fetch), "write_bytes": round(write), "launches_sampled": out["FETCH_SIZE"][1],
              "fetch_kib_min_max": [round(out["FETCH_SIZE"][2]), round(out["FETCH_SIZE"][3])],
              "formula": "2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)",
              "mfma_busy": busy}
    return round(fetch + write), detail


def cpu_baseline(work):
    """BASELINE config[0] (C1): unconditional DDIM-50 (eta=1), batch 2, DiTRotary_XL_8, on the host cores -- the torch-CPU
    restatement (oracle/dit_torch.py; same ATen operator set as the reference on CPU), three steps of the chain, median.
    Timed with torch's default thread count (what the reference would use) and with 16 threads (a batch-2 forward is
    weight-streaming work that many-core hosts oversubscribe); the faster of the two is the value, both are listed."""
    from oracle import diffusion_np as odf, dit_torch as odt
    sd = odt.to_torch({k: v.detach().cpu() for k, v in work.model.state_dict().items()})
    S = odf.Schedule(1000, "linear", "ddim50")

    def model(xx, tt):
        return odt.dit_forward(sd, xx, tt, None, depth=28, heads=16)

    def run(threads):
        torch.set_num_threads(threads)
        g = torch.Generator().manual_seed(0)
        x = torch.randn(2, 4, 128, 16, generator=g)
        times = []
        for i in (49, 48, 47, 46):                   # the first one also pages the 2.7 GB of weights in: not timed
            t = torch.full((2,), i, dtype=torch.int64)
            t0 = time.perf_counter()
            x, _ = odt.ddim_step(S, model, x, t, torch.randn(x.shape, generator=g), eta=1.0)
            times.append(time.perf_counter() - t0)
        return sorted(times[1:])[1], times[1:]
    default = torch.get_num_threads()
    tried = {}
    try:
        for th_ in sorted({default, min(16, default)}, reverse=True):
            tried[th_] = run(th_)
    finally:
        torch.set_num_threads(default)
    cores = min(tried, key=lambda k: tried[k][0])
    med, steps = tried[cores]
    return {"value": round(1.0 / med, 4), "unit": "steps/s", "cores": int(cores), "kind": "port",
            "sample": f"BASELINE C1: 3 DDIM-50 steps at batch 2 (DiTRotary_XL_8, torch CPU fp32 restatement), median {med:.2f} s/step with "
                      f"{cores} threads (steps: {', '.join(f'{v:.2f}' for v in steps)} s); "
                      + "; ".join(f"{k} threads: {v[0]:.2f} s/step" for k, v in tried.items())}


def uint8_flip_record(device):
    """tests/golden/e2e_ddim50_xl28.npz (the reference's 50-step B=2 chain): decoded uint8 roll vs the reference's."""
    from functools import partial
    from guided_diffusion.condition_functions import model_fn
    from guided_diffusion.gaussian_diffusion import _decode
    from guided_diffusion.midi_util import decode_sample_for_midi
    from rgm import synth
    from taming.models.klvae_pedal import AutoencoderKL
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "e2e_ddim50_xl28.npz")))
    m = build_eps_model(3, device)
    vae = AutoencoderKL()
    vae.load_state_dict(synth.vae_state_dict(2, device=device), strict=False)
    vae = vae.to(device).eval()
    d = make_diffusion("ddim50")
    d.batch_shard = d.scg_shard = False          # rank 0 runs this chain alone: no collectives
    rng = np.random.RandomState(701)
    q = [rng.randn(2, 4, 128, 16).astype(np.float32) for _ in range(51)]
    d.noise_fn = lambda shape, dev: torch.from_numpy(q.pop(0)).to(dev)
    lat = d.ddim_sample_loop(partial(model_fn, model=m, num_classes=3, class_cond=True, cfg=False, w=0.), (2, 4, 128, 16),
                             clip_denoised=False, model_kwargs={"y": torch.from_numpy(g["y"]).to(device)}, device=device, eta=1.0)
    err = float((lat.cpu().numpy().astype(np.float64) - g["latent"]).__abs__().max() / np.abs(g["latent"]).max())
    from rgm import native as R
    u8 = decode_sample_for_midi(lat, embed_model=vae, scale_factor=1.2465, threshold=-0.95).cpu().numpy()     # exact-fp32 decode (default)
    u8_loop = decode_sample_for_midi(lat, embed_model=vae, scale_factor=1.2465, threshold=-0.95, exact=False).cpu().numpy()
    with R.gemm_precision_scope("fp32"):
        roll = _decode(lat, vae, scale_factor=1.2465).cpu().numpy().astype(np.float64).transpose(0, 2, 3, 1)
    bad = u8 != g["u8"]
    qv = (roll[bad] + 1.0) * 63.5
    dist = np.minimum(np.abs(qv - np.round(qv)) / 63.5, np.abs(roll[bad] + 0.95)) if bad.any() else np.zeros(1)
    return {"mismatches": int(bad.sum()), "of": int(bad.size), "not_boundary_adjacent_1e-4": int((dist >= 1e-4).sum()),
            "latent_rel_err": float(f"{err:.3g}"),
            "final_decode": "exact fp32 MFMA (midi_util.FINAL_DECODE_EXACT), the 50 steps in the headline arithmetic",
            "mismatches_with_decode_in_loop_arithmetic": int((u8_loop != g["u8"]).sum())}


def eps_networks(work):
    from guided_diffusion.dit import DiTRotary
    return [v for v in vars(work).values() if isinstance(v, DiTRotary)]


def time_steps(work, steps, world, dist):
    """One timed region of exactly `steps` steps: barrier + synchronise on both sides, MAX over ranks.  Conditioning rows computed ahead
    (guided_diffusion/dit.py cond_hint: one pass over the adaLN weights serves the next 32 (t, y) rows of the schedule) are dropped first:
    every pass the region's steps need is paid INSIDE the region (at K = 20 one pass per region, where a 50-step chain pays two)."""
    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for m in eps_networks(work):
        m._ahead = None
    fence()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        work.step()
    e1.record()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=work.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, e0.elapsed_time(e1)


def scg_record(device, world, rank, dist, steps=5, warmup=2):
    """The sharded SCG step (north star): B = 4, n = 16, candidates partitioned over the ranks, one all-gather per step."""
    from rgm import scg_shard, batch_shard
    work = SCGWorkload(device, 4)
    ag = {"events": [], "row_events": []}
    real_gather, real_rows = scg_shard.gather_totals, batch_shard.gather_rows

    def timed(fn, key):
        def wrapper(arg):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn(arg)
            b.record()
            ag[key].append((a, b))
            return out
        return wrapper
    # the two collectives of a sharded search step: the eps (+ guidance gradient) rows of the x_t forward (batch_shard.gather_rows,
    # 32 KiB per sample) and the (n/R, B) table of rule log-probs (scg_shard.gather_totals, 64 B per sample)
    scg_shard.gather_totals = timed(real_gather, "events")
    batch_shard.gather_rows = timed(real_rows, "row_events")
    try:
        for _ in range(warmup):
            work.step()
        ag["events"].clear()
        ag["row_events"].clear()
        dt, _ = time_steps(work, steps, world, dist)
        torch.cuda.synchronize()
        ag_us = [1e3 * a.elapsed_time(b) for a, b in ag["events"]]
        row_us = [1e3 * a.elapsed_time(b) for a, b in ag["row_events"]]
        same = True
        if world > 1:
            mine = work.d.last_scg["max_ind"].to(torch.int64).contiguous()
            allm = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allm, mine)
            same = all(bool(torch.equal(allm[0], m)) for m in allm)
            tot = work.d.last_scg["total_log_prob"].contiguous()
            allt = [torch.empty_like(tot) for _ in range(world)]
            dist.all_gather(allt, tot)
            same = same and all(bool(torch.equal(allt[0], m)) for m in allt)
    finally:
        scg_shard.gather_totals, batch_shard.gather_rows = real_gather, real_rows
    return {"workload": work.name, "scaling": "strong", "ranks": world, "candidates_per_rank": 16 // world if 16 % world == 0 else 16,
            "steps": steps, "ms_per_step": round(1e3 * dt / steps, 3), "steps_per_s": round(steps / dt, 4),
            "allgather_us_per_step": round(float(np.median(ag_us)), 1) if ag_us else None,
            "allgather_logprob_table_us": round(float(np.median(ag_us)), 1) if ag_us else None,
            "allgather_eps_rows_us": round(float(np.median(row_us)), 1) if row_us else None,
            "same_winners_on_every_rank": bool(same),
            "algorithmic_tflops": round(work.flop_per_step * steps / dt / 1e12, 2)}


def c3_sharded_record(device, world, dist, steps=5, warmup=2):
    """BASELINE config[2] shape (classifier guidance, '250' chain, batch 32) as ONE chain whose batch is sharded over the ranks
    (rgm/batch_shard.py: rank r computes rows [r*B/R, (r+1)*B/R), one all-gather of x_{t-1} per step): strong scaling."""
    work = C3Workload(device, 32)
    work.d.batch_shard = True
    for _ in range(warmup):
        work.step()
    dt, _ = time_steps(work, steps, world, dist)
    mine = work.x.contiguous()
    allx = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allx, mine)
    same = all(bool(torch.equal(allx[0], m)) for m in allx)
    return {"workload": work.name + " [one chain, batch rows sharded over the ranks]", "scaling": "strong", "ranks": world,
            "rows_per_rank": 32 // world if 32 % world == 0 else 32, "steps": steps, "ms_per_step": round(1e3 * dt / steps, 3),
            "steps_per_s": round(steps / dt, 4), "same_latents_on_every_rank": bool(same)}


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: re-execute through torch.distributed.run, one rank per GPU, on 127.0.0.1."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("RGM_BENCH_ONE_DEVICE") != "1":   # (one-device plumbing mode: every rank on cuda:0 over gloo)
        sys.exit(f"bench.py: --gpus {n} requested but this box has {have} HIP device(s)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: spawning", " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "scg", "long", "dps_rule"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only (skip fp32_exact, scg, uint8_flips, cpu_baseline)")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)   # internal: the profiled child of measure_traffic
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-situ rocprofv3 --pmc passes (roofline.traffic from profiles/traffic.json)")
    ap.add_argument("--simulate-ranks", type=int, default=0,
                    help="--workload scg on ONE GPU: time the per-rank work of an R-GPU run (this process scores candidates "
                         "[0, n/R) like rank 0 would, the all-gather is a local stand-in): an estimate of the sharded step "
                         "time without the fabric; marked 'simulated' in the JSON, never the headline metric")
    ap.add_argument("--precision", default=os.environ.get("RGM_BENCH_PRECISION", "bf16x3_presplit"), choices=["fp32", "bf16x3", "bf16x3_presplit"],
                    help="GEMM arithmetic of the headline run: bf16x3 split on pre-split operands (default; fp32-grade: 2.5e-6 latent "
                         "error on the 50-step golden, the whole parity suite runs in all three modes) or exact fp32 MFMA")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)                                   # does not return
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    assert world == args.gpus, f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)"
    # RGM_BENCH_ONE_DEVICE=1 (plumbing test on a 1-GPU box only): every rank uses cuda:0 and the collectives run over gloo --
    # exercises the N > 1 control flow, never a measurement (the JSON says so)
    one_dev = os.environ.get("RGM_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        world = dist.get_world_size()                            # n_gpus = the RCCL world that actually runs

    from rgm import native as R
    R.set_gemm_precision(args.precision)
    torch.manual_seed(0)
    batch = args.batch or {"c2": 16, "c3": 32, "scg": 4, "long": 2, "dps_rule": 16}[args.workload]   # C5 as SURVEY 8(d) defines it: B = 2 (--batch 1: one long sample)
    work = {"c2": C2Workload, "c3": C3Workload, "scg": SCGWorkload, "long": LongWorkload,
            "dps_rule": DPSRuleWorkload}[args.workload](device, batch)
    if args.simulate_ranks > 1:
        assert args.workload in ("scg", "long") and world == 1, "--simulate-ranks is a single-GPU SCG experiment"
        from rgm import scg_shard
        Rn = args.simulate_ranks
        scg_shard.partition = lambda n, world_size=None, rank=None: (0, n // Rn, True) if n % Rn == 0 else (0, n, False)
        scg_shard.gather_totals = lambda local: local.repeat(Rn, 1)     # same table shape and selection work as the real all-gather
        from rgm import batch_shard                                     # the x_t forward of a search step: this rank's rows only
        batch_shard.partition_rows = lambda B, ws=None, r=None: ((0, B // Rn) if B % Rn == 0 else ((0, 1) if Rn % B == 0 else None))
        batch_shard.partition_roles = lambda B, ws=None, r=None: batch_shard._orig_partition_roles(B, Rn, 0)   # rank 0: an eps row (the longer role)
        batch_shard.gather_rows = lambda ts: [t.repeat((Rn,) + (1,) * (t.dim() - 1)) for t in ts]   # stand-in: shapes, not values
        batch_shard.window_ranks = lambda: Rn                          # B = 1 (C5): the collage windows of the x_t forward, rank 0's share
        batch_shard.window_world = lambda: (Rn, 0) if batch_shard.window_shard_on() else (1, 0)
        batch_shard.reduce_windows = lambda t: t                       # stand-in for the all-reduce of the window eps
    if args.workload in ("c2", "c3", "dps_rule"):
        work.d.batch_shard = False            # the headline runs one independent chain per GPU (weak scaling): nothing to shard
    for _ in range(args.warmup):
        work.step()
    if args.traffic_child:                                       # under rocprofv3 --pmc: just the steps, no timing, no JSON
        for _ in range(args.steps):
            work.step()
        torch.cuda.synchronize()
        return
    for m in eps_networks(work):
        m.ahead_passes = 0
    regions = [time_steps(work, args.steps, world, dist) for _ in range(max(1, args.repeats))]
    ahead_passes = sum(getattr(m, "ahead_passes", 0) for m in eps_networks(work))
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    dt, gpu_ms = regions[order[len(order) // 2]]
    try:                                                         # every rank runs it (SCG steps hold a collective)
        roof = roofline_pass(work)
    except Exception as e:                                       # never lose the headline number
        log("roofline pass failed:", repr(e))
        roof = None

    if roof is not None and rank == 0 and world == 1 and not args.no_traffic and not args.no_extras and args.simulate_ranks <= 1:
        try:
            tb, detail = measure_traffic(roof["kernel"], args)
        except Exception as e:
            tb, detail = None, repr(e)
        if tb is not None:
            mb = detail.pop("mfma_busy", None)
            roof["mfma_busy"] = mb["value"] if mb else None          # matrix-pipe duty of the dominant kernel (rocprofv3 --pmc, in situ)
            roof["mfma_busy_detail"] = mb
            roof["traffic"], roof["traffic_detail"] = tb, detail
            roof["traffic_over_algorithmic"] = round(tb / max(1.0, roof.get("algorithmic_bytes_per_launch", 0) or 1.0), 3) if roof.get("algorithmic_bytes_per_launch") else None
        else:
            roof["traffic_detail"] = {"method": "profiles/traffic.json lookup (in-situ measurement unavailable: %s)" % detail}
    if roof is not None and rank == 0 and world == 1 and not args.no_extras and args.simulate_ranks <= 1:
        try:
            pw = power_pass(work)
        except Exception as e:
            log("power pass failed:", repr(e))
            pw = None
        if pw:
            roof["power"] = pw
            # the same fraction against the MFMA peak at the clock the board sustained over the whole step (nominal peak: 2.4 GHz)
            roof["frac_at_sustained_clock"] = round(roof["frac"] * 2400.0 / max(pw["sclk_mhz_median"], 1.0), 4)
    extras = {}
    if args.workload == "c2" and not args.no_extras and args.simulate_ranks <= 1:
        def attempt(name, fn):
            try:
                extras[name] = fn()
            except Exception as e:
                log(f"{name} failed:", repr(e))
                extras[name] = None
        # exact-fp32 arithmetic of the same step (every rank: same model, other kernels)
        def fp32_exact():
            R.set_gemm_precision("fp32")
            try:
                for _ in range(2):
                    work.step()
                k = max(3, min(args.steps, 10))
                d32, _ = time_steps(work, k, world, dist)
                r32 = roofline_pass(work)
            finally:
                R.set_gemm_precision(args.precision)
            return {"dtype": "f32 (v_mfma_f32_32x32x2_f32)", "steps": k, "value": round(k * world / d32, 4), "unit": "steps/s",
                    "ms_per_step": round(1e3 * d32 / k, 3), "roofline": r32}
        attempt("fp32_exact", fp32_exact)
        attempt("scg", lambda: scg_record(device, world, rank, dist))
        if world > 1:
            attempt("c3_batch_sharded", lambda: c3_sharded_record(device, world, dist))
        if rank == 0:
            attempt("uint8_flips", lambda: uint8_flip_record(device))
            if world == 1 and not args.no_cpu_baseline:
                attempt("cpu_baseline", lambda: cpu_baseline(work))
    same_winners = None
    if world > 1 and args.workload in ("scg", "long"):
        # every rank must have picked the same candidates in the last step (identical gathered table -> identical first-argmax)
        mine = work.d.last_scg["max_ind"].to(torch.int64).contiguous()
        allm = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        tot = work.d.last_scg["total_log_prob"].contiguous()
        allt = [torch.empty_like(tot) for _ in range(world)]
        dist.all_gather(allt, tot)
        same_winners = all(bool(torch.equal(allm[0], m)) for m in allm) and all(bool(torch.equal(allt[0], m)) for m in allt)
    if rank == 0:
        sharded = args.workload in ("scg", "long")
        units = args.steps * (1 if sharded else world)           # SCG shards ONE chain; C2 runs one chain per GPU
        res = {
            "metric": "denoising steps/sec (whole node), DiTRotary_XL_8 4x128x16",
            "value": round(units / dt, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else f"f32 via {args.precision} split (3 bf16 MFMA per product, fp32 accumulate)",
            "data": "synthetic" if not one_dev else "synthetic [PLUMBING TEST: all ranks on one device over gloo -- not a measurement]",
            "config": {"workload": work.name + (f" [SIMULATED rank 0 of {args.simulate_ranks}: per-rank work only, no fabric]"
                                                if args.simulate_ranks > 1 else ""), "batch_per_gpu": batch, "sample_steps_per_s": round(units * batch / dt, 2),
                       "weights": "synthetic random-init (rgm.synth seed 1; adaLN/final layers re-randomised)",
                       "timing": f"median of {len(regions)} timed regions of {args.steps} steps",
                       "repeats_ms_per_step": [round(1e3 * r[0] / args.steps, 3) for r in regions],
                       "gpu_ms_per_step_events": round(gpu_ms / args.steps, 3),
                       # adaLN conditioning ahead of the step (guided_diffusion/dit.py): rows per pass over the adaLN weights (0 = every forward
                       # streams them itself) and the passes the timed regions paid -- each region starts without rows
                       "cond_ahead": {"rows_per_pass": __import__("guided_diffusion.dit", fromlist=["COND_AHEAD"]).COND_AHEAD,
                                      "weight_passes_in_timed_regions": ahead_passes, "timed_steps": args.steps * len(regions)},
                       # whole-step FLOPs over ONE simulated rank's time would read as several times the chip's peak: no figure there
                       "algorithmic_tflops": None if args.simulate_ranks > 1 else round(work.flop_per_step * units / dt / 1e12, 2)},
        }
        if roof is not None and res["config"]["algorithmic_tflops"] is not None:
            # the number the north star's ">= 40 % on the DiT forward" is about: the step's algorithmic FLOPs over the timed
            # region, against the peak of the arithmetic the step ran in (per GPU)
            wpeak = F32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else BF16X3_EQUIV_PEAK_TFLOPS
            roof["whole_step_tflops"] = round(res["config"]["algorithmic_tflops"] / world, 2)
            roof["whole_step_frac"] = round(res["config"]["algorithmic_tflops"] / world / wpeak, 4)
        if same_winners is not None:
            res["config"]["same_winners_on_every_rank"] = bool(same_winners)
        if "uint8_flips" in extras:
            res["config"]["uint8_flips"] = extras.pop("uint8_flips")
        res["roofline"] = roof
        for k in ("fp32_exact", "scg", "c3_batch_sharded", "cpu_baseline"):
            if k in extras:
                res[k] = extras[k]
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
