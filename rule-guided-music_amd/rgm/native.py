"""Load librgm_hip.so and declare the C ABI of include/rgm.h."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RGM_LIB_PATH", os.path.join(_HERE, "librgm_hip.so"))   # override: kernel experiments only


class RgmError(RuntimeError):
    pass


class DitCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("depth", "hidden", "heads", "patch", "in_ch", "out_ch", "width",
                                         "n_embed", "kind", "n_out", "max_tokens")]


def _load():
    if not os.path.exists(LIB_PATH):
        raise RgmError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(or `make -C rule-guided-music_amd/csrc`).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    sig = {
        "rgm_version": (C.c_int, []),
        "rgm_last_error": (C.c_char_p, []),
        "rgm_dit_create": (C.c_int, [C.POINTER(DitCfg), C.POINTER(vp)]),
        "rgm_dit_destroy": (None, [vp]),
        "rgm_dit_set_param": (C.c_int, [vp, C.c_char_p, vp, C.POINTER(C.c_int64), i32]),
        "rgm_dit_missing_params": (C.c_int, [vp]),
        "rgm_dit_workspace_bytes": (sz, [vp, i32, i32]),
        "rgm_dit_forward": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, vp, sz, vp]),
        "rgm_dit_cond_rows": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, sz, vp]),
        "rgm_dit_forward_cond": (C.c_int, [vp, vp, vp, vp, i32, vp, i32, i32, vp, sz, vp]),
        "rgm_dit_classify": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, vp, sz, vp]),
        "rgm_gemm": (C.c_int, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, i32, f32, vp, i32, i32, vp, i32, vp]),
        "rgm_gemm_tile": (C.c_int, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, i32, i32, vp]),
        "rgm_adaln_stream": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp]),
        "rgm_layernorm_modulate": (C.c_int, [vp, vp, i32, i32, f32, vp, vp, vp, vp, i32, i32, vp]),
        "rgm_rotary_attention": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "rgm_rotary_attention_lse": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "rgm_randn": (C.c_int, [vp, C.c_int64, C.c_uint64, C.c_uint64, vp]),
        "rgm_ddpm_step": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, vp]),
        "rgm_ddpm_step_learned": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, vp]),
        "rgm_ddpm_step_learned_g": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, vp]),
        "rgm_scg_rebuild_g": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint64, vp, i32, i32, i32, i32, i32, vp]),
        "rgm_ddim_step": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, f32, vp, vp, vp, i32, i32, vp]),
        "rgm_scg_candidates": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp]),
        "rgm_xstart_from_eps": (C.c_int, [vp, vp, vp, vp, f32, vp, i32, i32, vp]),
        "rgm_edit_replace_eps": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, vp]),
        "rgm_scg_select": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp]),
        "rgm_scg_rebuild": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint64, vp, i32, i32, i32, i32, i32, vp]),
        "rgm_vae_create": (C.c_int, [C.POINTER(vp)]),
        "rgm_vae_destroy": (None, [vp]),
        "rgm_vae_set_param": (C.c_int, [vp, C.c_char_p, vp, C.POINTER(C.c_int64), i32]),
        "rgm_vae_has_param": (C.c_int, [vp, C.c_char_p]),
        "rgm_vae_missing_params": (C.c_int, [vp]),
        "rgm_vae_encoder_missing_params": (C.c_int, [vp]),
        "rgm_vae_encode": (C.c_int, [vp, vp, vp, i32, vp, sz, vp]),
        "rgm_vae_workspace_bytes": (sz, [vp, i32]),
        "rgm_vae_decode": (C.c_int, [vp, vp, vp, i32, vp, sz, vp]),
        "rgm_vae_decode_latent": (C.c_int, [vp, vp, f32, vp, vp, f32, i32, i32, vp, sz, vp]),
        "rgm_vae_enable_grad": (C.c_int, [vp]),
        "rgm_vae_grad_workspace_bytes": (sz, [vp, i32]),
        "rgm_vae_decode_latent_save": (C.c_int, [vp, vp, f32, vp, i32, i32, vp, sz, vp]),
        "rgm_vae_decode_latent_vjp": (C.c_int, [vp, vp, f32, vp, i32, i32, vp, sz, vp]),
        "rgm_quantise_roll": (C.c_int, [vp, vp, i32, i32, f32, vp]),
        "rgm_rule_pitch_hist": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
        "rgm_rule_pitch_hist_vag": (C.c_int, [vp, vp, f32, vp, vp, vp, vp, i32, i32, i32, vp]),
        "rgm_rule_note_density": (C.c_int, [vp, vp, i32, i32, i32, i32, f32, vp]),
        "rgm_rule_chord_quantise": (C.c_int, [vp, vp, i32, i32, i32, vp]),
        "rgm_bucketize": (C.c_int, [vp, vp, i32, vp, i32, vp]),
        "rgm_row_loss": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
        "rgm_collage_split": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
        "rgm_collage_merge": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
        "rgm_dit_grad_workspace_bytes": (sz, [vp, i32, i32]),
        "rgm_dit_cls_value_and_grad": (C.c_int, [vp, vp, vp, vp, i32, f32, vp, vp, i32, i32, vp, sz, vp]),
        "rgm_dit_enable_grad": (C.c_int, [vp]),
        "rgm_dit_vjp": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, sz, vp]),
        "rgm_rotary_attention_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "rgm_split_rows": (C.c_int, [vp, vp, C.c_int64, i32, vp]),
        "rgm_gemm_split": (C.c_int, [vp, vp, vp, i32, i32, i32, vp, i32, i32, i32, vp]),
        "rgm_gemm_scratch_bytes": (sz, [i32, i32]),
        "rgm_set_big_tiles": (C.c_int, [i32, i32]),
        "rgm_set_fuse_reduce_ln": (C.c_int, [i32]),
        "rgm_set_adaln_overlap": (C.c_int, [i32]),
        "rgm_set_dit_halves": (C.c_int, [i32, vp]),
        "rgm_set_attn_pairs": (C.c_int, [i32]),
        "rgm_set_gn_fuse": (C.c_int, [i32, vp]),
        "rgm_gn_fused_launches": (C.c_longlong, []),
        "rgm_gn_fallback_tiles": (C.c_longlong, [i32]),
        "rgm_split_dtype": (C.c_int, []),
        "rgm_set_attn_split": (C.c_int, [i32]),
        "rgm_fused_reduce_ln_launches": (C.c_longlong, []),
        "rgm_gemm_split_ws": (C.c_int, [vp, vp, vp, i32, i32, i32, vp, i32, i32, i32, vp, sz, vp]),
        "rgm_gemm_split_epi": (C.c_int, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, i32, f32, vp, i32, i32, vp, i32, i32, i32, vp, sz, vp]),
        "rgm_split_rows_ld": (C.c_int, [vp, i32, vp, i32, C.c_int64, i32, vp]),
        "rgm_gemm_split_ld": (C.c_int, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, i32, i32, i32, vp]),
        "rgm_set_gemm_precision": (C.c_int, [i32]),
        "rgm_get_gemm_precision": (C.c_int, []),
        "rgm_prof_enable": (C.c_int, [i32]),
        "rgm_prof_reset": (C.c_int, []),
        "rgm_prof_report": (C.c_int, [i32, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "rgm_prof_bytes": (C.c_double, [i32]),
        "rgm_prof_dump": (C.c_int, [i32, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "rgm_gemm2_dbg": (C.c_int, [i32, C.POINTER(C.c_longlong)]),
        "rgm_gemm144_dbg": (C.c_int, [i32, C.POINTER(C.c_longlong)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)            # AttributeError here == header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return L


class _Lazy:
    """Defer dlopen to first use so CPU-only tooling can import the package metadata."""
    _lib = None

    def __getattr__(self, name):
        if _Lazy._lib is None:
            _Lazy._lib = _load()
        return getattr(_Lazy._lib, name)


lib = _Lazy()
PRECISIONS = {"fp32": 0, "bf16x3": 1, "bf16x3_presplit": 2}


def set_gemm_precision(name):
    """Process-wide GEMM arithmetic: 'fp32' (exact v_mfma_f32_32x32x2_f32; the library's start-up value), 'bf16x3' (operands split
    hi + lo into two bf16 while staged, 3 bf16 MFMAs per product, ~2^-16 relative per product) or 'bf16x3_presplit' (same numerics,
    operands split once by their producer, LDS-DMA staging; the default of bench.py and the CLIs)."""
    check(lib.rgm_set_gemm_precision(PRECISIONS[name]))


class gemm_precision_scope:
    """`with gemm_precision_scope("fp32"): ...` -- the arithmetic for the launches enqueued inside the block, the previous one
    restored behind it.  The precision is read on the host when a launch is enqueued (never by a kernel in flight), so the scope is
    exact for a rank's single sampling thread.  Used for the ONE decode per run whose output is an integer piano roll
    (guided_diffusion/midi_util.decode_sample_for_midi, reference midi_util.py:42-64): the exact-fp32 decoder there, bf16x3 everywhere
    else (SCG's inner decodes included)."""

    def __init__(self, name):
        self.name = name
        self.prev = None

    def __enter__(self):
        self.prev = int(lib.rgm_get_gemm_precision())
        if self.name is not None:
            check(lib.rgm_set_gemm_precision(PRECISIONS[self.name]))
        return self

    def __exit__(self, *exc):
        check(lib.rgm_set_gemm_precision(self.prev))
        return False


def check(status):
    if status != 0:
        raise RgmError(f"librgm_hip status {status}: {lib.rgm_last_error().decode()}")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RgmError("rgm ops need tensors on a HIP device (no CPU fallback in the product path)")


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise RgmError("rgm ops need contiguous tensors")
    return C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
