#!/usr/bin/env python3
"""Writes the sampling / editing YAML configs the reference ships (scripts/configs/**: the tables and ablations of the paper) into
rule-guided-music_amd/scripts/configs/ -- same file names and schema (the CLI contract), this repository's layout.

Every VALUE (guidance method, cond_fn lists, classifier names / scales / paths, SCG weights, schedules, respacing, dc, edit) is
read from the reference tree in THIS container and re-emitted unchanged -- tests/test_host_logic.py checks the shipped tree
against tests/golden/ref_configs.json (the reference's YAMLs, parsed).  The ONE intended difference:
  * `target_rules: Null` (targets drawn from a dataset batch) is not supported by the sampling CLI here (no dataset loader):
    explicit example targets are written instead (edit configs keep Null = "the source's own rule", which edit.py supports).
Chord entries are kept (the CLIs skip them with a warning unless --chord_backend registers an analyser); `edit.source: dataset`
is kept (edit.py falls back to its synthetic source with a warning: the test-set loader is out of scope).
Files of this repository's own (cond_demo/demo_long.yml, cond_demo/demo2_plain.yml, cond_table/no_guidance/uncond_ddim50.yml)
are not in the reference tree and are left alone.
Run:  python tools/make_configs.py          (needs /root/reference; the outputs are committed)"""
import glob
import os
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/scripts/configs"
OUT = os.path.join(ROOT, "rule-guided-music_amd", "scripts", "configs")

EXAMPLE = {   # explicit targets where the reference draws them from its dataset
    "pitch_hist": [0.2, 0., 0.1, 0., 0.15, 0.15, 0., 0.2, 0., 0.1, 0., 0.1],
    "vertical_nd": [1.5, 3., 4.5, 3., 1.5, 3., 4.5, 3.],
    "horizontal_nd": [10., 15., 20., 15., 10., 15., 20., 15.],
    "chord_progression": [1, 1, 4, 4, 5, 5, 1, 1],
    "chord_progression_pixel": [1, 1, 4, 4, 5, 5, 1, 1],
}
NOTE = {
    "pitch_hist": "weights of the 12 pitch classes (normalised to sum 1 by the loader)",
    "vertical_nd": "notes sounding per frame, one value per 1.28 s window",
    "horizontal_nd": "onsets per window (divided by the rule's horizontal scale by the loader)",
    "chord_progression": "Roman-numeral degree per window (needs the music21 chord analyser: register_chord_backend)",
    "chord_progression_pixel": "Roman-numeral degree per window (needs the music21 chord analyser: register_chord_backend)",
}


def flow(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if v is None:
        return "null"
    if isinstance(v, float):
        return repr(v).rstrip("0") if "." in repr(v) else repr(v)
    if isinstance(v, (int, str)):
        return str(v)
    if isinstance(v, list):
        return "[" + ", ".join(flow(x) for x in v) + "]"
    if isinstance(v, dict):
        return "{" + ", ".join(f"{k}: {flow(x)}" for k, x in v.items()) + "}"
    raise TypeError(type(v))


def describe(rel, d):
    g = d["guidance"]
    what = []
    if g.get("method") == "classifier_guidance":
        what.append("classifier guidance on x_t")
    elif g.get("method") == "dps":
        what.append("DPS guidance through " + ("classifiers on the x0 estimate" if g.get("nn") else "the rule on the decoded roll"))
    if g.get("scg") or "scg" in d:
        what.append(f"SCG with {d['scg']['num_samples']} candidates per step")
    if not what:
        what.append("no guidance (rule losses of the unguided model)")
    samp = d["sampling"]
    how = f"DDIM ({samp.get('timestep_respacing')})" if samp.get("use_ddim") else "DDPM"
    if samp.get("t_end"):
        how += f", stopped early at t = {samp['t_end']}"
    if g.get("schedule"):
        how += f"; guided for {g.get('t_end', 0)} <= t < {g.get('t_start')}" + (f", every {g['interval']} steps" if g.get("interval", 1) != 1 else "")
    space = "latent diffusion + VAE decode" if g.get("vae") else "no VAE in the loop"
    return f"# {rel}: " + " + ".join(what) + f"; {how}; {space}."


def emit(rel, d):
    lines = [describe(rel, d)]
    if "latent/" in rel and not d["guidance"].get("vae"):
        lines.append("# Pixel-space ablation of the paper (the eps-network generates the roll itself): needs a pixel-space checkpoint and\n"
                     "# --image_size / --in_channels to match; the native kernels are validated at the latent shape only.")
    samp = dict(d["sampling"])
    order = ["use_ddim", "timestep_respacing", "diff_collage", "t_end"]
    lines.append("sampling: " + flow({k: samp[k] for k in order if k in samp}))
    if "dc" in d:
        lines.append("dc: " + flow(d["dc"]))
    if "edit" in d:
        e = dict(d["edit"])
        lines.append("edit: " + flow(e) + "   # source: dataset (the reference's test-set excerpt; here: the synthetic source), a .npy roll or a MIDI file")
    lines.append("")
    g = dict(d["guidance"])
    lines.append("guidance:")
    for k in ("method", "nn", "vae", "scg", "beam", "step_size", "schedule", "t_start", "t_end", "interval", "dc"):
        if k in g:
            lines.append(f"  {k}: {flow(g[k])}")
    c = g.get("cond_fn")
    if c is None:
        lines.append("  cond_fn: null")
    else:
        lines.append("  cond_fn:")
        for k in ("fns", "rule_names", "classifier_scales"):
            lines.append(f"    {k}: {flow(c[k])}")
        if "classifiers" in c:
            cl = dict(c["classifiers"])
            lines.append("    classifiers:" + ("" if "names" in cl else "   # no `names` in the reference's pixel-space configs: the CLI infers them from fns"))
            for k in ("names", "num_classes", "paths"):
                if k in cl:
                    lines.append(f"      {k}: {flow(cl[k])}")
    if "scg" in d:
        lines.append("")
        lines.append("scg: " + flow(d["scg"]) + "   # candidates per step, then one weight per rule's log-probability (default 1)")
    lines.append("")
    lines.append("target_rules:")
    for k, v in d["target_rules"].items():
        if v is None and "edit" not in d:
            base = k.replace("_hr_2", "").replace("_hr_1", "")
            lines.append(f"  {k}: {flow(EXAMPLE[base])}   # {NOTE[base]}")
        else:
            lines.append(f"  {k}: {flow(v)}" + ("   # null = keep the source's own value" if v is None else ""))
    return "\n".join(lines) + "\n"


def main():
    if not os.path.isdir(REF):
        sys.exit("needs the reference checkout at /root/reference")
    n = 0
    for f in sorted(glob.glob(os.path.join(REF, "**", "*.yml"), recursive=True)):
        rel = os.path.relpath(f, REF)
        dst = os.path.join(OUT, rel)
        d = yaml.safe_load(open(f))
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w") as o:
            o.write(emit(rel, d) + "# (generated by tools/make_configs.py)\n")
        n += 1
    print(f"wrote {n} configs under {OUT}")


if __name__ == "__main__":
    main()
