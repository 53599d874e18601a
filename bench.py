#!/usr/bin/env python3
"""bench.py -- denoising steps/s of the rule-guided sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Metric (BASELINE.json): denoising steps/sec, whole node, DiTRotary_XL_8 on 4x128x16 latents.
Workload at N=1 = BASELINE config[1] ("C2"): unconditional DDIM-50 (eta=1) on a batch of 16 latents, one
"step" = one ddim_sample call over that batch (eps-network forward + fused step update + Philox noise).
For N>1 every rank runs the same batch-16 chain on its own GPU (the path shards over samples; no data-path
collective), scaling = weak, value = N*K / max-over-ranks time.  Other workloads (--workload scg) time the
SCG branch-and-select step (config[3] shape) with the RCCL log-prob all-gather.

One JSON line on stdout (rank 0) per the driver contract, plus
  roofline     : the dominant kernel (fp32-MFMA GEMM, 128x128 tile) -- algorithmic 2MNK FLOPs of its launches
                 divided by their HIP-event durations measured live on the launch stream (separate short pass
                 with rgm_prof_enable, same workload), against the 157.3 TFLOP/s f32 matrix peak;
  cpu_baseline : the numpy oracle (oracle/, kind "port") timed on this box's host cores on a bounded sample.
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
VAE_GFLOP_PER_TILE = 114.48
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
    """One guided SCG step (config[3] shape): B=4, n=16 candidates, DiT-XL + VAE decode + 2 rules, DDPM chain."""
    name = "C4 SCG guided DDPM step, DiTRotary_XL_8 + KL-VAE decode + pitch_hist/note_density, B=4, n=16 (sharded over GPUs)"

    def __init__(self, device, batch):
        from functools import partial
        from types import SimpleNamespace
        from rgm import synth
        from guided_diffusion.condition_functions import model_fn
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
        self.guid = SimpleNamespace(schedule=True, t_start=750, t_end=0, interval=1, method="no_guidance")
        self.scg = {"num_samples": 16, "pitch_hist": 40., "note_density": 1.}
        self.k = 0
        self.flop_per_step = batch * ((1 + 16) * DIT_GFLOP_PER_SAMPLE + 16 * 8 * VAE_GFLOP_PER_TILE) * 1e9

    def step(self):
        i = 700 - (self.k % 600)
        self.k += 1
        t = torch.full((self.B,), i, dtype=torch.int64, device=self.device)
        self.d._t_host = i
        out = self.d.p_sample(self.fn, self.x, t, clip_denoised=False, model_kwargs=self.kw, embed_model=self.vae,
                              scale_factor=1.2465, guidance_kwargs=self.guid, scg_kwargs=self.scg)
        self.d._t_host = None
        self.x = out["sample"]


class LongWorkload(SCGWorkload):
    """BASELINE config[4]: a 40.96 s sequence -- linear DiffCollage over 7 windows of 128 latent rows with overlap 64
    (latent 4 x 512 x 16) -- with SCG (n = 16) scoring rule(decode(x0)) on the 4096-frame roll."""
    name = ("C5 DiffCollage (linear, 7 windows, overlap 64) + SCG guided DDPM step, DiTRotary_XL_8 + KL-VAE decode of 32 squares "
            "per candidate + pitch_hist/note_density, n=16 (sharded over GPUs)")

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
        windows = 7 + 6                       # 7 full windows + 6 overlap halves evaluated by the collage
        self.flop_per_step = batch * ((1 + 16) * windows * DIT_GFLOP_PER_SAMPLE + 16 * (H // 16) * VAE_GFLOP_PER_TILE) * 1e9


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
    R.check(R.lib.rgm_prof_reset())
    R.check(R.lib.rgm_prof_enable(1))
    for _ in range(steps):
        work.step()
    torch.cuda.synchronize()
    R.check(R.lib.rgm_prof_enable(0))
    rows = {}
    for kid in range(1, 140):
        n, ms, fl = C.c_int(), C.c_double(), C.c_double()
        R.check(R.lib.rgm_prof_report(kid, C.byref(n), C.byref(ms), C.byref(fl)))
        if n.value:
            rows[kid] = dict(launches=n.value, ms=ms.value, flops=fl.value)
    R.check(R.lib.rgm_prof_reset())
    # kernel ids (csrc/gemm.hip, gemm2.hip): < 40 gemm_kernel<BM,BN,WM,WN,ALOAD,PREC> = tile + 10*ALOAD + 20*PREC;
    # >= 40 gemm2_kernel<BM,BN,WM,WN,ALOAD,NSTAGE,0,PIPE> = 40 + tile + 10*ALOAD (tiles: gemm2_launch)
    tiles = {1: "128,128,2,2", 2: "128,64,2,2", 3: "64,64,2,2", 4: "32,128,1,4", 5: "256,128,4,2"}
    g2 = {1: (1, 3, 0), 2: (2, 3, 0), 3: (3, 3, 0), 5: (5, 3, 0), 21: (1, 2, 1), 22: (2, 2, 1),
          43: (1, 2, 3), 44: (2, 2, 3), 45: (5, 2, 3), 46: (3, 3, 3), 51: (1, 3, 4), 52: (2, 3, 4)}

    def kname(k):
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
            "share_of_gemm_time": round(r["ms"] / all_ms, 3),
            "all_gemm_tflops": round(sum(v["flops"] for v in rows.values()) / (all_ms * 1e-3) / 1e12, 2)}


def cpu_baseline(work, batch):
    """The numpy oracle (a port; the reference itself cannot travel) on the host cores: DDIM steps of the same
    chain on a bounded sample."""
    from oracle import diffusion_np as odf, dit_np as odit
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        cores = os.cpu_count() or 1
    sd = {k: v.detach().cpu().numpy() for k, v in work.model.state_dict().items()}
    S = odf.Schedule(1000, "linear", "ddim50")
    b = min(batch, 4)
    rng = np.random.RandomState(0)
    x = rng.randn(b, 4, 128, 16).astype(np.float32)

    def model(xx, tt, **kw):
        return odit.dit_forward(sd, xx, tt, None, depth=28, heads=16)
    n_steps, t0 = 0, time.perf_counter()
    for i in (49, 48, 47):
        t = np.full((b,), i, dtype=np.int64)
        x = odf.ddim_sample(S, model, x, t, rng.randn(*x.shape).astype(np.float32), eta=1.0)["sample"]
        n_steps += 1
        if time.perf_counter() - t0 > 20:
            break
    dt = time.perf_counter() - t0
    # one step of batch `batch` costs batch/b times a step of batch b (GEMM-bound, linear in batch)
    return {"value": round(n_steps / dt * b / batch, 4), "unit": "steps/s", "cores": int(cores), "kind": "port",
            "sample": f"{n_steps} DDIM steps of the same chain at batch {b} with the numpy oracle ({dt:.1f} s), scaled to batch {batch}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "scg", "long", "dps_rule"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--simulate-ranks", type=int, default=0,
                    help="--workload scg on ONE GPU: time the per-rank work of an R-GPU run (this process scores candidates "
                         "[0, n/R) like rank 0 would, the all-gather is a local stand-in): an estimate of the sharded step "
                         "time without the fabric; marked 'simulated' in the JSON, never the headline metric")
    ap.add_argument("--precision", default=os.environ.get("RGM_BENCH_PRECISION", "bf16x3_presplit"), choices=["fp32", "bf16x3", "bf16x3_presplit"],
                    help="GEMM arithmetic: bf16x3 split (default; fp32-grade: 2.5e-6 latent error on the 50-step golden, "
                         "parity suite runs in both modes) or exact fp32 MFMA")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from rgm import native as R
    R.set_gemm_precision(args.precision)
    torch.manual_seed(0)
    batch = args.batch or {"c2": 16, "c3": 32, "scg": 4, "long": 1, "dps_rule": 16}[args.workload]
    work = {"c2": C2Workload, "c3": C3Workload, "scg": SCGWorkload, "long": LongWorkload,
            "dps_rule": DPSRuleWorkload}[args.workload](device, batch)
    if args.simulate_ranks > 1:
        assert args.workload in ("scg", "long") and world == 1, "--simulate-ranks is a single-GPU SCG experiment"
        from rgm import scg_shard
        Rn = args.simulate_ranks
        scg_shard.partition = lambda n, world_size=None, rank=None: (0, n // Rn, True) if n % Rn == 0 else (0, n, False)
        scg_shard.gather_totals = lambda local: local.repeat(Rn, 1)     # same table shape and selection work as the real all-gather
    for _ in range(args.warmup):
        work.step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        work.step()
    e1.record()
    fence()
    dt = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    try:                                                         # every rank runs it (SCG steps hold a collective)
        roof = roofline_pass(work)
    except Exception as e:                                       # never lose the headline number
        log("roofline pass failed:", repr(e))
        roof = None
    if rank == 0:
        sharded = args.workload in ("scg", "long")
        units = args.steps * (1 if sharded else world)           # SCG shards ONE chain; C2 runs one chain per GPU
        res = {
            "metric": "denoising steps/sec (whole node), DiTRotary_XL_8 4x128x16",
            "value": round(units / dt, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else f"f32 via {args.precision} split (3 bf16 MFMA per product, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": work.name + (f" [SIMULATED rank 0 of {args.simulate_ranks}: per-rank work only, no fabric]"
                                                if args.simulate_ranks > 1 else ""), "batch_per_gpu": batch, "sample_steps_per_s": round(units * batch / dt, 2),
                       "weights": "synthetic random-init (rgm.synth seed 1; adaLN/final layers re-randomised)",
                       "gpu_ms_per_step_events": round(gpu_ms / args.steps, 3),
                       "algorithmic_tflops": round(work.flop_per_step * units / dt / 1e12, 2)},
        }
        res["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline and args.workload == "c2":
            try:
                res["cpu_baseline"] = cpu_baseline(work, batch)
            except Exception as e:
                log("cpu baseline failed:", repr(e))
                res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
