"""ctypes binding of libfourm_hip.so (include/fourm_hip.h).  No torch types cross this boundary:
only device pointers, sizes and the raw hipStream_t.

The library is built in-tree by ``ml-4m_amd/build_ext.py`` (``__graft_entry__.build()``).  Its
absence is a hard error: there is no CPU or eager-PyTorch fallback for the hot path.
"""
import ctypes as C
import os

FM_MAX_MODS = 24
_LIB_PATH = os.environ.get("FOURM_HIP_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_lib", "libfourm_hip.so")
# (FOURM_HIP_LIB: a kernel-lab build of the same ABI, e.g. tools/r05_attn_variants.sh; the product always loads the in-tree library)


class FourmHipUnavailable(ImportError):
    pass


def _load():
    if not os.path.exists(_LIB_PATH):
        raise FourmHipUnavailable(
            f"{_LIB_PATH} is missing - build the gfx950 kernels first: python ml-4m_amd/build_ext.py "
            "(the 4M hot path has no fallback implementation)")
    return C.CDLL(_LIB_PATH)


lib = _load()
lib.fm_last_error.restype = C.c_char_p
lib.fm_abi_version.restype = C.c_int
ABI_VERSION = 8
if lib.fm_abi_version() != ABI_VERSION:
    raise FourmHipUnavailable(f"libfourm_hip.so ABI {lib.fm_abi_version()} != expected {ABI_VERSION}; rebuild")

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# enums (keep in sync with the header; tests/test_model_cpu.py::test_ctypes_mirrors_match_the_header_layout cross-checks them, the struct layouts and the prototypes against the header)
EPI_BF16, EPI_GELU, EPI_RESIDUAL, EPI_SWIGLU, EPI_F32, EPI_TANH, EPI_SWIGLU_BWD, EPI_GELU_BWD = 0, 1, 2, 3, 4, 5, 6, 7
MASK_NONE, MASK_KEYPAD, MASK_DECODER, MASK_DENSE = 0, 1, 2, 3
KIND_TOK, KIND_PATCH, KIND_SEQ, KIND_SEQ_EMB = 0, 1, 2, 3
LOSS_MOD, LOSS_TOKEN = 0, 1


class GemmGroup(C.Structure):
    _fields_ = [("W", vp), ("out", vp), ("N", i32), ("K", i32), ("ldw", i32), ("pad_", i32)]


class GemmNTArgs(C.Structure):
    _fields_ = [("W", vp), ("W2", vp), ("X", vp), ("out", vp), ("out2", vp), ("res", vp), ("bias", vp), ("bias2", vp),
                ("M", i32), ("N", i32), ("K", i32), ("ldw", i32), ("ldx", i32), ("ldo", i32), ("ldo2", i32), ("ldr", i32),
                ("Hp", i32), ("epilogue", i32), ("groups", vp), ("tile_group", vp), ("max_N", i32), ("pad_", i32),
                ("m_dev", vp), ("row0_dev", vp), ("splitk_ws", vp), ("splitk_ws_bytes", i64),
                ("conv_C", i32), ("conv_H", i32), ("conv_W", i32), ("conv_Ho", i32), ("conv_Wo", i32), ("conv_stride", i32), ("conv_up", i32), ("conv_pad_", i32)]


class GemmTNArgs(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("out", vp), ("R", i32), ("N", i32), ("K", i32), ("lda", i32), ("ldb", i32),
                ("ldo", i32), ("a_cols", i32), ("b_cols", i32), ("splits", i32), ("force_tr", i32), ("groups", vp),
                ("seg_start", vp), ("seg_count", vp), ("n_groups", i32), ("max_N", i32), ("max_R", i32), ("pad_", i32)]


class GemmTNJob(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("out", vp), ("R", i32), ("N", i32), ("K", i32), ("lda", i32), ("ldb", i32),
                ("ldo", i32), ("a_cols", i32), ("b_cols", i32)]


TN_MAX_JOBS = 16


class AttnArgs(C.Structure):
    _fields_ = [("Q", vp), ("K", vp), ("V", vp), ("O", vp), ("stat_m", vp), ("stat_l", vp),
                ("ldq", i32), ("ldk", i32), ("ldv", i32), ("ldo", i32),
                ("B", i32), ("H", i32), ("Nq", i32), ("Nk", i32), ("head_dim", i32), ("mask_kind", i32),
                ("scale", f32), ("causal", i32),
                ("kpad", vp), ("cs", vp), ("modq", vp), ("modk", vp), ("dense", vp),
                ("dO", vp), ("dQ", vp), ("dK", vp), ("dV", vp),
                ("lddo", i32), ("lddq", i32), ("lddk", i32), ("lddv", i32), ("force_tr", i32), ("kv_batch_rows", i32), ("zero_attn", i32)]


class ModDesc(C.Structure):
    _fields_ = [("ids", vp), ("mask", vp), ("dam", vp), ("table", vp), ("pos", vp), ("mod_emb", vp), ("proj_bias", vp),
                ("L", i32), ("kind", i32), ("ids_are_i64", i32), ("mod_id", i32), ("max_len", i32), ("shifted", i32),
                ("mask_stride", i32), ("id_stride", i32), ("patch", i32), ("channels", i32), ("grid_w", i32),
                ("orig_dim", i32), ("head_index", i32), ("pad_", i32)]


class ShadowDesc(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("ld_src", i32), ("ld_dst", i32), ("rows", i32), ("cols", i32),
                ("transpose", i32), ("tile_start", i32), ("col_scale", vp), ("dst_f32", i32), ("reserved", i32)]


FOLD_MAX_JOBS = 48


class FoldGradJob(C.Structure):
    _fields_ = [("dWp", vp), ("W", vp), ("gamma", vp), ("gW", vp), ("ggamma", vp), ("rows", i32), ("cols", i32), ("ld_dwp", i32), ("reserved", i32)]


class GemmF32Args(C.Structure):
    _fields_ = [("X", vp), ("W", vp), ("W2", vp), ("out", vp), ("out2", vp), ("res", vp), ("bias", vp), ("bias2", vp),
                ("sxm", i64), ("sxk", i64), ("swn", i64), ("swk", i64),
                ("M", i32), ("N", i32), ("K", i32), ("ldo", i32), ("ldo2", i32), ("ldr", i32), ("Hp", i32), ("epilogue", i32),
                ("accumulate", i32), ("max_N", i32), ("seg_rows", i32), ("n_groups", i32),
                ("groups", vp), ("tile_group", vp), ("seg_start", vp), ("seg_count", vp)]


class SpanMaskArgs(C.Structure):
    _fields_ = [("ids", vp), ("len", vp), ("unit", vp), ("noise", vp), ("keep_prob", vp), ("r_choice", vp), ("input_budget", vp), ("target_budget", vp),
                ("sentinel_ids", vp), ("emb", vp), ("emb_out", vp), ("src", vp), ("tensor", vp), ("input_mask", vp), ("target_mask", vp),
                ("decoder_attention_mask", vp), ("tries", vp),
                ("B", i32), ("ld_ids", i32), ("T", i32), ("ld_noise", i32), ("n_sentinels", i32), ("max_tokens", i32), ("vocab_offset", i32), ("pad_id", i32),
                ("emb_rows", i32), ("emb_dim", i32)]


class AdamWJob(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("dst_plain", vp), ("dst_t", vp),
                ("rows", i32), ("cols", i32), ("ld_plain", i32), ("ld_t", i32), ("tile_start", i32), ("pad_", i32)]


class SelectDesc(C.Structure):
    _fields_ = [("mods", ModDesc * FM_MAX_MODS),
                ("n_mods", i32), ("batch", i32), ("dim", i32), ("n_keep", i32), ("n_reg", i32), ("total_len", i32),
                ("is_decoder", i32), ("raw", i32),
                ("reg_tokens", vp), ("mask_token", vp),
                ("tokens", vp), ("emb", vp), ("x0", vp), ("out_mask", vp), ("out_mod", vp), ("slot_mod", vp),
                ("slot_src", vp), ("slot_pos", vp), ("target_ids", vp), ("out_cs", vp), ("out_mod_pre", vp),
                ("out_mod_index", vp), ("patch_rows", vp), ("seqemb_rows", vp), ("patch_ld", i32), ("seqemb_ld", i32),
                ("rows_f32", i32), ("pad2_", i32)]


class EmbedBwdMod(C.Structure):
    _fields_ = [("d_table", vp), ("d_pos", vp), ("d_mod_emb", vp), ("d_proj_bias", vp),
                ("kind", i32), ("has_padding_idx", i32), ("padding_idx", i32), ("pad_", i32)]


class EmbedBwdDesc(C.Structure):
    _fields_ = [("mods", EmbedBwdMod * FM_MAX_MODS), ("dx", vp), ("slot_mod", vp), ("slot_src", vp), ("slot_pos", vp),
                ("d_mask_token", vp), ("d_reg_tokens", vp),
                ("n_mods", i32), ("batch", i32), ("dim", i32), ("Nt", i32), ("lddx", i32), ("is_decoder", i32)]


def _sig(name, *argtypes):
    fn = getattr(lib, name)
    fn.argtypes = list(argtypes)
    fn.restype = C.c_int
    return fn


P = C.POINTER
gemm_nt = _sig("fm_gemm_nt", P(GemmNTArgs), vp)
gemm_tn = _sig("fm_gemm_tn", P(GemmTNArgs), vp)
gemm_tn_multi = _sig("fm_gemm_tn_multi", P(GemmTNJob), C.c_int, vp)
layernorm_fwd = _sig("fm_layernorm_fwd", vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, f32, vp)
layernorm_fwd_res = _sig("fm_layernorm_fwd_res", vp, i32, vp, i32, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, f32, vp)
layernorm_bwd = _sig("fm_layernorm_bwd", vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, i32, i32, vp)
layernorm_bwd_h = _sig("fm_layernorm_bwd_h", vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, vp, i32, i32, vp)
attn_fwd = _sig("fm_attn_fwd", P(AttnArgs), vp)
attn_bwd = _sig("fm_attn_bwd", P(AttnArgs), vp)
select_embed = _sig("fm_select_embed", P(SelectDesc), vp)
embed_bwd = _sig("fm_embed_bwd", P(EmbedBwdDesc), vp)
dense_decoder_mask = _sig("fm_dense_decoder_mask", vp, vp, vp, i32, i32, i32, i32, i32, vp)
segment_rows = _sig("fm_segment_rows", vp, i32, i32, vp, vp, vp, vp, vp, i32, vp)
gather_rows = _sig("fm_gather_rows", vp, i32, vp, vp, i32, i32, i32, vp)
cross_entropy = _sig("fm_cross_entropy", vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp)
swiglu_bwd = _sig("fm_swiglu_bwd", vp, i32, vp, i32, vp, i32, i32, i32, i32, vp)
gelu_bwd = _sig("fm_gelu_bwd", vp, i32, vp, i32, vp, i32, i32, i32, i32, vp)
cast_pad = _sig("fm_cast_pad", vp, i32, vp, i32, i32, i32, vp)
transpose_cast_pad = _sig("fm_transpose_cast_pad", vp, i32, vp, i32, i32, i32, i32, vp)
colsum = _sig("fm_colsum", vp, i32, vp, i32, i32, vp)
shadow_refresh = _sig("fm_shadow_refresh", vp, i32, i32, vp)
fold_colscale_grad = _sig("fm_fold_colscale_grad", vp, i32, vp)
headnorm_fwd = _sig("fm_headnorm_fwd", vp, i32, vp, vp, vp, i32, vp, i32, i32, C.c_float, vp)
headnorm_bwd = _sig("fm_headnorm_bwd", vp, i32, vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, vp)
f32_to_bf16 = _sig("fm_f32_to_bf16", vp, vp, i64, vp)
bf16_to_f32_scaled = _sig("fm_bf16_to_f32_scaled", vp, vp, i64, f32, vp)
add_bf16_f32 = _sig("fm_add_bf16_f32", vp, vp, vp, i64, vp)
lib.fm_set_reserved_cus.argtypes = [C.c_int]
lib.fm_get_reserved_cus.restype = C.c_int
lib.fm_get_reserved_cus.argtypes = []
adamw = _sig("fm_adamw", vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, vp, vp, vp, vp)
adamw_shadow = _sig("fm_adamw_shadow", vp, i32, i32, f32, f32, f32, f32, f32, i64, vp, vp, vp, vp)
sumsq = _sig("fm_sumsq", vp, i64, vp, vp)
clip_coef = _sig("fm_clip_coef", vp, f32, vp, vp, vp)
sample_tokens = _sig("fm_sample_tokens", vp, i32, i32, i32, i32, f32, i32, f32, vp, vp, vp, vp)
maskgit_commit = _sig("fm_maskgit_commit", vp, vp, vp, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp)
# fp32 verification path
gemm_f32 = _sig("fm_gemm_f32", P(GemmF32Args), vp)
attn_f32_fwd = _sig("fm_attn_f32_fwd", P(AttnArgs), vp)
attn_f32_bwd = _sig("fm_attn_f32_bwd", P(AttnArgs), vp)
layernorm_bwd_f32 = _sig("fm_layernorm_bwd_f32", vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, i32, i32, vp)
headnorm_f32_fwd = _sig("fm_headnorm_f32_fwd", vp, i32, vp, vp, vp, i32, vp, i32, i32, C.c_float, vp)
headnorm_f32_bwd = _sig("fm_headnorm_f32_bwd", vp, i32, vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, vp)
swiglu_bwd_f32 = _sig("fm_swiglu_bwd_f32", vp, i32, vp, i32, vp, i32, i32, i32, i32, vp)
gelu_bwd_f32 = _sig("fm_gelu_bwd_f32", vp, i32, vp, i32, vp, i32, i32, i32, vp)
colsum_f32 = _sig("fm_colsum_f32", vp, i32, vp, i32, i32, vp)
cross_entropy_f32 = _sig("fm_cross_entropy_f32", vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp)
vq_patchify = _sig("fm_vq_patchify", vp, vp, i32, i32, i32, i32, i32, i32, vp)
l2norm_rows = _sig("fm_l2norm_rows", vp, i32, vp, i32, i32, i32, vp)
vq_assign = _sig("fm_vq_assign", vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp)
lib.fm_set_tn_transpose_read.argtypes = [C.c_int]
lib.fm_set_gemm_tn_config.argtypes = [C.c_int]
lib.fm_set_attn_transpose_read.argtypes = [C.c_int]

lib.fm_set_gemm_nt_config.argtypes = [C.c_int]
lib.fm_lab_set.argtypes = [C.c_int, C.c_int]      # lab knobs (header, "lab" section)
lib.fm_lab_set.restype = None
vq_code_stats = _sig("fm_vq_code_stats", vp, i32, vp, i32, i32, i32, vp, vp, vp)
vq_ema_update = _sig("fm_vq_ema_update", vp, vp, vp, vp, i32, i32, f32, vp)
vq_code_bias = _sig("fm_vq_code_bias", vp, i32, i32, vp, vp)
vq_assign_bias = _sig("fm_vq_assign_bias", vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp)
vq_code_stats_raw = _sig("fm_vq_code_stats_raw", vp, i32, vp, i32, i32, i32, vp, vp, vp)
vq_ema_update_euclid = _sig("fm_vq_ema_update_euclid", vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, vp)
vq_unpatchify = _sig("fm_vq_unpatchify", vp, i32, vp, i32, i32, i32, i32, i32, vp)
vq_latent_grad = _sig("fm_vq_latent_grad", vp, i32, vp, vp, vp, i32, vp, f32, vp, i32, vp, i32, i32, vp)
tanh_bwd_f32 = _sig("fm_tanh_bwd_f32", vp, vp, vp, i32, i32, i32, vp)
embed_rows_f32 = _sig("fm_embed_rows_f32", vp, vp, vp, i32, i32, i32, vp)
vq_patchify_ex = _sig("fm_vq_patchify_ex", vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp)
vq_cls_emb_bwd = _sig("fm_vq_cls_emb_bwd", vp, i32, vp, vp, i32, i32, i32, i32, i32, vp)
vq_latent_grad_normalized = _sig("fm_vq_latent_grad_normalized", vp, i32, vp, vp, vp, i32, vp, f32, vp, i32, vp, i32, i32, vp)
scale_rows_bf16 = _sig("fm_scale_rows_bf16", vp, i32, vp, i32, i32, i32, vp)
image_mask = _sig("fm_image_mask", vp, vp, vp, i32, i32, vp, vp, vp, vp)
token_budgets = _sig("fm_token_budgets", vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp)
span_mask = _sig("fm_span_mask", P(SpanMaskArgs), vp)
guidance_combine = _sig("fm_guidance_combine", vp, i32, i32, vp, i32, i32, f32, vp, i32, i32, i32, i32, vp)
unpack_image_u8 = _sig("fm_unpack_image_u8", vp, vp, i32, i32, i32, i32, P(C.c_float), P(C.c_float), vp)
unpack_ids_u16 = _sig("fm_unpack_ids_u16", vp, vp, i64, vp)
unpack_mask_bits = _sig("fm_unpack_mask_bits", vp, vp, i32, i32, vp)
decoder_attention_from_target = _sig("fm_decoder_attention_from_target", vp, vp, i32, i32, vp)
split3_bf16 = _sig("fm_split3_bf16", vp, i32, vp, i32, i32, i32, i32, i32, vp)
unet_im2col = _sig("fm_unet_im2col", vp, i32, i32, vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp)
groupnorm_nhwc = _sig("fm_groupnorm_nhwc", vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, f32, i32, vp)
add_bf16 = _sig("fm_add_bf16", vp, i32, vp, i32, vp, i32, i64, i32, vp)
silu_f32_to_bf16 = _sig("fm_silu_f32_to_bf16", vp, vp, i64, vp)
timestep_embedding = _sig("fm_timestep_embedding", vp, vp, i32, i32, i32, f32, vp)
unet_attention = _sig("fm_unet_attention", vp, i32, vp, i32, i32, i32, i32, i32, vp)
diffusion_x0 = _sig("fm_diffusion_x0", vp, vp, f32, f32, vp, i64, vp)
quantile_abs = _sig("fm_quantile_abs", vp, i32, i64, f32, vp, vp)
diffusion_step = _sig("fm_diffusion_step", vp, vp, f32, f32, vp, vp, vp, f32, f32, f32, f32, vp, vp, i32, i64, vp)
EXPORTS = ["fm_split3_bf16", "fm_unet_im2col", "fm_groupnorm_nhwc", "fm_add_bf16", "fm_silu_f32_to_bf16", "fm_timestep_embedding", "fm_unet_attention", "fm_diffusion_x0", "fm_quantile_abs",
           "fm_diffusion_step", "fm_unpack_image_u8", "fm_unpack_ids_u16", "fm_unpack_mask_bits", "fm_decoder_attention_from_target", "fm_guidance_combine", "fm_image_mask", "fm_token_budgets", "fm_span_mask", "fm_vq_code_stats", "fm_vq_ema_update", "fm_vq_code_bias", "fm_vq_assign_bias", "fm_vq_code_stats_raw", "fm_vq_ema_update_euclid", "fm_vq_unpatchify", "fm_vq_latent_grad", "fm_tanh_bwd_f32", "fm_embed_rows_f32", "fm_vq_patchify_ex", "fm_vq_cls_emb_bwd", "fm_vq_latent_grad_normalized", "fm_abi_version", "fm_last_error", "fm_gemm_nt", "fm_set_gemm_nt_config", "fm_get_gemm_nt_config", "fm_set_reserved_cus", "fm_get_reserved_cus", "fm_bf16_to_f32_scaled", "fm_add_bf16_f32", "fm_scale_rows_bf16", "fm_gemm_tn", "fm_gemm_tn_multi", "fm_set_gemm_tn_config", "fm_get_gemm_tn_config", "fm_set_tn_transpose_read",
           "fm_get_tn_transpose_read", "fm_layernorm_fwd", "fm_layernorm_fwd_res", "fm_layernorm_bwd", "fm_layernorm_bwd_h", "fm_headnorm_fwd", "fm_headnorm_bwd", "fm_attn_fwd", "fm_attn_bwd",
           "fm_set_attn_transpose_read", "fm_get_attn_transpose_read", "fm_select_embed", "fm_embed_bwd",
           "fm_dense_decoder_mask", "fm_segment_rows", "fm_gather_rows", "fm_cross_entropy", "fm_swiglu_bwd",
           "fm_gelu_bwd", "fm_cast_pad", "fm_transpose_cast_pad", "fm_shadow_refresh", "fm_fold_colscale_grad", "fm_colsum", "fm_f32_to_bf16", "fm_adamw", "fm_adamw_shadow",
           "fm_sumsq", "fm_clip_coef", "fm_vq_patchify", "fm_l2norm_rows", "fm_vq_assign",
           "fm_sample_tokens", "fm_maskgit_commit", "fm_gemm_f32", "fm_attn_f32_fwd", "fm_attn_f32_bwd", "fm_layernorm_bwd_f32", "fm_headnorm_f32_fwd", "fm_headnorm_f32_bwd",
           "fm_swiglu_bwd_f32", "fm_gelu_bwd_f32", "fm_colsum_f32", "fm_cross_entropy_f32", "fm_lab_set"]


def check(rc: int):
    if rc != 0:
        raise RuntimeError(f"libfourm_hip: {lib.fm_last_error().decode()} (rc={rc})")
