"""-m gpu: scripts/sample_rule.py end to end with synthetic weights (no checkpoints exist offline) on short chains:
flags, YAML handling, sampler, decode, rule report and CSV outputs -- the four shipped config families."""
import importlib.util
import os

import numpy as np
import pandas as pd
import pytest

from conftest import PKG

pytestmark = pytest.mark.gpu
CFG = os.path.join(PKG, "scripts", "configs")


def _cli():
    spec = importlib.util.spec_from_file_location("sample_rule_cli", os.path.join(PKG, "scripts", "sample_rule.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


COMMON = ["--model", "DiTRotary_B_8", "--image_size", "128", "16", "--in_channels", "4", "--scale_factor", "1.2465",
          "--class_cond", "True", "--num_classes", "3", "--class_label", "1", "--synthetic_weights", "True", "--progress", "False"]


@pytest.mark.parametrize("cfg,extra,rules", [
    ("cond_table/no_guidance/uncond_ddim50.yml", [], ["pitch_hist"]),
    ("cond_demo/demo2.yml", ["--diffusion_steps", "25"], ["pitch_hist", "note_density"]),
    ("cond_table/single/classifier/nd.yml", ["--diffusion_steps", "25"], ["note_density"]),
    ("cond_table/all/scg_classifier_all.yml", ["--diffusion_steps", "25"], ["pitch_hist", "note_density"]),
    ("cond_demo/demo_long.yml", ["--diffusion_steps", "24"], ["pitch_hist", "note_density"]),
    ("cond_table/single/dps_nn/nd.yml", ["--diffusion_steps", "25"], ["note_density"]),
    ("cond_table/single/dps_rule/pitch.yml", ["--diffusion_steps", "24"], ["pitch_hist"]),
    ("cond_table/single/dps_rule/nd.yml", ["--diffusion_steps", "24"], ["note_density"]),
    # from the generated part of the tree (tools/make_configs.py): SCG on one rule, DDIM-respaced SCG, scheduled guidance
    # every 5 steps, early stopping, the unguided baseline, classifier + SCG with a different candidate count
    ("cond_table/single/scg/pitch.yml", ["--diffusion_steps", "24"], ["pitch_hist"]),
    ("cond_table/abla/sampling/ddim/ddim25.yml", [], ["note_density"]),
    ("cond_table/abla/sampling/ddpm/every5.yml", ["--diffusion_steps", "24"], ["note_density"]),
    ("cond_table/abla/sampling/ddpmes/s750_400.yml", ["--diffusion_steps", "430"], ["note_density"]),   # stops at t = 400: 30 steps
    ("cond_table/no_guidance/nd.yml", ["--diffusion_steps", "24"], ["note_density"]),
    ("cond_table/abla/combine/nd_scg_cls_num4.yml", ["--diffusion_steps", "24"], ["note_density"]),
    ("cond_table/single/dps_nn/pitch.yml", ["--diffusion_steps", "24"], ["pitch_hist"]),
])
def test_sample_rule_cli(tmp_path, monkeypatch, cfg, extra, rules):
    monkeypatch.chdir(tmp_path)
    cli = _cli()
    has_chord = "chord" in open(os.path.join(CFG, cfg)).read()
    if has_chord:                                 # a chord config without an analyser is an error unless the run opts out of those rules
        with pytest.raises(RuntimeError, match="skip_chord_rules"):
            cli.main(["--config_path", os.path.join(CFG, cfg), "--batch_size", "2", "--num_samples", "2"] + COMMON + extra)
        extra = extra + ["--skip_chord_rules", "True"]
    res = cli.main(["--config_path", os.path.join(CFG, cfg), "--batch_size", "2", "--num_samples", "2"] + COMMON + extra)
    out_dir = os.path.join("loggings", cli.output_dir_for(os.path.join(CFG, cfg), 1)) + ("_nochord" if has_chord else "")
    import json
    meta = json.load(open(os.path.join(out_dir, "run_metadata.json")))
    assert bool(meta["dropped_rules"]) == has_chord and meta["synthetic_weights"] is True
    df = pd.read_csv(os.path.join(out_dir, "results.csv"))
    assert len(df) == 2 and len(res) == 2
    for r in rules:
        assert {f"{r}.target_rule", f"{r}.gen_rule", f"{r}.loss"} <= set(df.columns)
        assert np.isfinite(df[f"{r}.loss"]).all()
    assert os.path.exists(os.path.join(out_dir, "summary.csv"))
    rolls = sorted(f for f in os.listdir(out_dir) if f.startswith("sample_") and f.endswith(".npy"))
    assert rolls == ["sample_0_y_1.npy", "sample_1_y_1.npy"]
    roll = np.load(os.path.join(out_dir, rolls[0]))
    T = 4096 if "long" in cfg else 1024
    assert roll.shape == (3, 128, T) and roll.dtype == np.uint8 and roll.max() <= 127


def test_sample_rule_cli_takes_null_targets_from_a_dataset_batch_file(tmp_path, monkeypatch):
    """`target_rules: Null` (every cond_table/** YAML of the reference; its targets are the rules of a dataset batch, reference
    scripts/sample_rule.py:147-168): --targets_npz hands that batch over (`gt` rolls) and the CLI extracts every rule with
    _extract_rule like the reference; vertical_nd / horizontal_nd collapse into note_density.  Also the one-array-per-rule form."""
    import torch
    monkeypatch.chdir(tmp_path)
    cli = _cli()
    cfg = os.path.join(str(tmp_path), "configs", "cond_table", "all", "null_targets.yml")
    os.makedirs(os.path.dirname(cfg))
    src = open(os.path.join(CFG, "cond_table", "no_guidance", "nd.yml")).read()
    text = src[:src.index("target_rules:")] + "target_rules: {pitch_hist: Null, vertical_nd: Null, horizontal_nd: Null}\n"
    open(cfg, "w").write(text)
    rng = np.random.RandomState(4)
    gt = -np.ones((2, 3, 128, 1024), dtype=np.float32)
    for b in range(2):
        for _ in range(40):
            pch, st, ln = rng.randint(30, 100), rng.randint(0, 1000), rng.randint(8, 100)
            gt[b, 0, pch, st:st + ln] = rng.uniform(-0.2, 1.0)
            gt[b, 1, pch, st] = 1.0
    npz = os.path.join(str(tmp_path), "batch.npz")
    np.savez(npz, gt=gt)
    args = ["--config_path", cfg, "--batch_size", "2", "--num_samples", "2", "--diffusion_steps", "24"] + COMMON
    with pytest.raises(NotImplementedError, match="targets_npz"):
        cli.main(args)
    res = cli.main(args + ["--targets_npz", npz])
    want = {k: cli._extract_rule(k, torch.from_numpy(gt).cuda()).cpu().numpy() for k in ("pitch_hist", "note_density")}
    for k, w in want.items():
        got = np.array([np.asarray(v, dtype=np.float64) for v in res[f"{k}.target_rule"]])
        assert got.shape == w.shape and np.allclose(got, w, atol=1e-6), k
        assert np.isfinite(res[f"{k}.loss"]).all()
    assert not np.allclose(want["pitch_hist"][0], want["pitch_hist"][1])          # per-sample targets, not one broadcast row
    npz2 = os.path.join(str(tmp_path), "rules.npz")
    np.savez(npz2, pitch_hist=want["pitch_hist"], note_density=want["note_density"][0])
    res2 = cli.main(args + ["--targets_npz", npz2])
    got = np.array([np.asarray(v, dtype=np.float64) for v in res2["note_density.target_rule"]])
    assert np.allclose(got, np.stack([want["note_density"][0]] * 2), atol=1e-6)


def test_edit_cli_keeps_the_fixed_part_and_rewrites_the_excerpt(tmp_path, monkeypatch):
    """scripts/edit.py end to end (synthetic weights, synthetic source, 24-step chain, noise_level from the YAML clipped to
    the chain): the latent rows outside [l_start, l_end) are the encoded source (replacement conditioning), the
    report covers the edited excerpt only."""
    import torch
    monkeypatch.chdir(tmp_path)
    spec = importlib.util.spec_from_file_location("edit_cli", os.path.join(PKG, "scripts", "edit.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    cfg_src = os.path.join(CFG, "edit", "nd_scg_given_target.yml")
    cfg = os.path.join(str(tmp_path), "configs", "edit", "nd_short.yml")
    os.makedirs(os.path.dirname(cfg))
    open(cfg, "w").write(open(cfg_src).read().replace("noise_level: 500", "noise_level: 12"))
    with pytest.raises(RuntimeError, match="allow_synthetic_source"):          # `source: dataset` is never replaced silently
        cli.main(["--config_path", cfg, "--batch_size", "2", "--num_samples", "2", "--diffusion_steps", "24"] + COMMON)
    res, sample = cli.main(["--config_path", cfg, "--batch_size", "2", "--num_samples", "2", "--diffusion_steps", "24",
                            "--allow_synthetic_source", "True"] + COMMON)
    assert len(res) == 2 and {"note_density.loss", "note_density.orig_rule"} <= set(res.columns)
    assert np.isfinite(res["note_density.loss"]).all()
    out_dir = os.path.join("loggings", "edit_demo", "edit", "nd_short_cls_1_synthsrc")
    import json
    assert json.load(open(os.path.join(out_dir, "run_metadata.json")))["source_substituted_by_synthetic"] is True
    assert os.path.exists(os.path.join(out_dir, "results.csv")) and os.path.exists(os.path.join(out_dir, "gt", "sample_0_y_1.npy"))
    assert sample.shape == (2, 128, 1024, 3) and sample.dtype == torch.uint8


def test_edit_cli_reads_a_midi_source(tmp_path, monkeypatch):
    """edit.source = a MIDI file: the built-in SMF reader -> (3,128,T) roll -> padded with background -> encoded; the run's gt/
    directory holds the re-written source as .midi again (default writer)."""
    from conftest import load_golden
    from music_rule_guidance.piano_roll_to_chord import piano_roll_to_pretty_midi
    monkeypatch.chdir(tmp_path)
    spec = importlib.util.spec_from_file_location("edit_cli", os.path.join(PKG, "scripts", "edit.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    src = os.path.join(str(tmp_path), "source.midi")
    piano_roll_to_pretty_midi(load_golden("midi_events")["r3.roll"].astype(np.float32), fs=100).write(src)
    cfg_src = os.path.join(CFG, "edit", "nd_scg_given_target.yml")
    cfg = os.path.join(str(tmp_path), "configs", "edit", "nd_midi.yml")
    os.makedirs(os.path.dirname(cfg))
    text = open(cfg_src).read().replace("noise_level: 500", "noise_level: 12").replace("source: dataset", f"source: {src}")
    open(cfg, "w").write(text)
    res, sample = cli.main(["--config_path", cfg, "--batch_size", "1", "--num_samples", "1", "--diffusion_steps", "24"] + COMMON)
    assert len(res) == 1 and np.isfinite(res["note_density.loss"]).all()
    gt_dir = os.path.join("loggings", "edit_demo", "edit", "nd_midi_cls_1", "gt")
    assert os.path.exists(os.path.join(gt_dir, "sample_0_y_1.midi")) and os.path.exists(os.path.join(gt_dir, "sample_0_y_1.npy"))
    gt = np.load(os.path.join(gt_dir, "sample_0_y_1.npy"))
    assert gt.shape == (3, 128, 1024) and (gt[0, :, 384:] == 0).all() and gt[0, :, :384].max() > 0


@pytest.mark.parametrize("flags", [["--cfg", "True", "--w", "4.", "--class_cond", "True"], ["--class_cond", "False", "--use_ddim", "True",
                                                                                       "--timestep_respacing", "ddim12"]])
def test_cfg_sample_cli(tmp_path, monkeypatch, flags):
    """scripts/cfg_sample.py (reference :26-127): unguided / classifier-free-guided batches -> uint8 rolls -> files of rank 0."""
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("OPENAI_LOGDIR", str(tmp_path / "log"))
    spec = importlib.util.spec_from_file_location("cfg_sample_cli", os.path.join(PKG, "scripts", "cfg_sample.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    arr = cli.main(["--model", "DiTRotary_B_8", "--image_size", "128", "16", "--in_channels", "4", "--scale_factor", "1.2465",
                    "--num_classes", "3", "--class_label", "2", "--synthetic_weights", "True", "--progress", "False",
                    "--batch_size", "2", "--num_samples", "3", "--diffusion_steps", "24", "--save_name", "_t"] + flags)
    assert arr.shape == (3, 3, 128, 1024) and arr.dtype == np.uint8 and arr.max() <= 127
    assert not np.array_equal(arr[0], arr[1])                                  # different noise per sample
    out = [os.path.join(dp, f) for dp, _, fs in os.walk(str(tmp_path)) for f in fs if f.startswith("sample_")]
    stems = sorted(os.path.basename(f) for f in out)
    if "--cfg" in flags:
        assert stems == ["sample_0_y_2.midi", "sample_0_y_2.npy", "sample_1_y_2.midi", "sample_1_y_2.npy", "sample_2_y_2.midi", "sample_2_y_2.npy"]
    else:
        assert stems == ["sample_0.midi", "sample_0.npy", "sample_1.midi", "sample_1.npy", "sample_2.midi", "sample_2.npy"]
    assert all("gen_cls_2_t" in f for f in out)
