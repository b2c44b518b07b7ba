"""ctypes binding of libvln_hip.so (C ABI: include/vln_hip.h).

The product path has NO fallback: if the HIP library is missing or a symbol is
absent, importing / calling fails loudly.  `python __graft_entry__.py` (or
`make -C curriculum-learning-for-vln_amd/csrc`) builds the library in-tree.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvln_hip.so")

i32, i64, u64, f32 = C.c_int, C.c_int64, C.c_uint64, C.c_float
ptr = C.c_void_p


class VlnError(RuntimeError):
    pass


class EnvDropDims(C.Structure):
    _fields_ = [(n, i32) for n in ("B", "L", "V", "C", "H", "IMG", "ANG", "AE", "wtype", "ctype")]


class DotStep(C.Structure):          # vln_dot_step
    _fields_ = [("ctx", ptr), ("vec", ptr), ("dots", ptr), ("S", i32)]


class SelectStep(C.Structure):       # vln_select_step
    _fields_ = [("src", ptr), ("index", ptr), ("out", ptr), ("C", i32)]


class WsumStep(C.Structure):         # vln_wsum_step
    _fields_ = [("ctx", ptr), ("w", ptr), ("out", ptr), ("S", i32), ("probs", ptr), ("target", ptr)]


class GatherRolloutStep(C.Structure):
    _fields_ = ([(n, ptr) for n in ("rows", "view_index", "crows", "cviews", "heading", "elevation", "out", "out_bf16", "cout", "cout_bf16")]
                + [("offset_pano", u64), ("offset_cand", u64)])


class GatherRide(C.Structure):         # vln_gather_ride
    _fields_ = ([("table", ptr), ("angle_table", ptr), ("steps", ptr)] + [(n, i32) for n in ("ttype", "T", "B", "V", "C", "IMG", "ANG", "pad_")]
                + [("seed", u64), ("p_feat", f32), ("padf_", f32), ("offset_base_dev", ptr)]
                + [("fetch_slots", ptr), ("fetch_seq", ptr), ("fetch_dst", ptr), ("fetch_offset", i64), ("fetch_bytes", i64), ("fetch_ring", i32), ("pad2_", i32)]
                + [("shadow_jobs", ptr), ("n_shadow_jobs", i32), ("pad3_", i32)])


class CatStep(C.Structure):
    _fields_ = [("probs", ptr), ("action", ptr), ("dlogits", ptr), ("C", i32)]


class CeStep(C.Structure):           # vln_ce_step
    _fields_ = [("logits", ptr), ("ld", i64), ("target", ptr), ("cand_mask", ptr), ("probs", ptr), ("dlogits", ptr), ("C", i32)]


CE_MAX_STEPS = 40                     # VLN_CE_MAX_STEPS


class MonitorLossStep(C.Structure):  # vln_monitor_loss_step
    _fields_ = [("logits", ptr), ("ld", i64), ("target", ptr), ("cand_mask", ptr), ("probs", ptr), ("progress", ptr), ("ldp", i64),
                ("start_dist", ptr), ("cur_dist", ptr), ("ended", ptr), ("prog_target", ptr), ("dlogits", ptr), ("dprogress", ptr), ("C", i32)]


MONITOR_LOSS_MAX_STEPS = 16           # VLN_MONITOR_LOSS_MAX_STEPS


class EnvDropWeights(C.Structure):
    _fields_ = [(n, ptr) for n in ("act_w", "act_b", "w_vin", "w_vin_t", "w_cat", "w_cat_t", "b_ih", "b_hh",
                                   "w_tin", "w_tin_t", "w_tout", "w_tout_t", "w_c", "w_c_t")] + [("f32_mask", i32), ("pad_", i32)]


class EnvDropStep(C.Structure):
    _fields_ = ([(n, ptr) for n in ("a_prev", "img", "cand", "img_lp", "cand_lp", "h_tilde_prev", "c0", "ctx",
                                    "ctx_lp", "ctx_mask", "logit", "h1", "c1", "h_tilde", "e", "xcat", "hq",
                                    "alpha_v", "gate_act", "tanh_c1", "tcat", "tt", "alpha_t", "htd", "a_stash")]
                + [("seed", u64), ("offset", u64), ("p_drop", f32), ("p_feat", f32), ("already_dropfeat", i32), ("lp_ready", i32),
                   ("ws", ptr), ("ws_floats", i64), ("offset_dev", ptr), ("offset_base_dev", ptr), ("defer_logits", i32),
                   ("pad_", i32)]
                + [(n, ptr) for n in ("g_table", "g_angle_table", "g_rows", "g_vidx", "g_crows", "g_cviews", "g_chead", "g_celev")]
                + [("g_ttype", i32), ("pad2_", i32), ("attn_sync", ptr), ("attn_sync_bytes", i64), ("chain", i32), ("pad3_", i32), ("kctx", ptr)]
                + [(n, ptr) for n in ("s_cand_mask", "s_action_in", "s_action_out", "s_action_host", "s_probs", "s_logp", "s_ent")]
                + [("s_seed", u64), ("s_offset", u64), ("s_offset_base_dev", ptr)])


class TickItem(C.Structure):          # vln_tick_item
    _fields_ = [("word", ptr), ("inc", u64), ("width", i32), ("pad_", i32)]


class ShadowJob(C.Structure):
    _fields_ = [("src", ptr), ("src2", ptr), ("dst", ptr), ("dst_t", ptr), ("ld_src", i64), ("ld_dst", i64), ("ld_dst_t", i64),
                ("N", i32), ("K", i32), ("out_type", i32), ("pad_", i32)]


class WgradJob(C.Structure):
    _fields_ = [("dy", ptr), ("x", ptr), ("dw", ptr), ("ld_dy", i64), ("ld_x", i64), ("ld_dw", i64),
                ("N", i32), ("K", i32), ("accumulate", i32), ("rows", i32)]


class ColsumJob(C.Structure):
    _fields_ = [("A", ptr), ("out1", ptr), ("out2", ptr), ("lda", i64), ("cols", i32), ("accumulate", i32)]


PARAM_JOBS_MAX = 8                    # VLN_PARAM_JOBS_MAX


class ParamJobs(C.Structure):
    _fields_ = [("w", WgradJob * PARAM_JOBS_MAX), ("c", ColsumJob * PARAM_JOBS_MAX), ("nw", i32), ("nc", i32), ("rows", i32),
                ("precision", i32)]


class EnvDropGrads(C.Structure):
    _fields_ = [(n, ptr) for n in ("dlogit", "dh1", "dc1", "dh_tilde", "dh_tilde_prev", "dc0", "dctx", "s_dtc",
                                   "s_dz", "s_dtt", "s_dgates", "s_dtv", "s_de", "s_dl", "s_dtcat", "dhtd_ext")]


class MonitorDims(C.Structure):
    _fields_ = [(n, i32) for n in ("B", "L", "C", "H", "M", "wtype")]


class MonitorWeights(C.Structure):
    _fields_ = [(n, ptr) for n in ("w_tin", "w_tin_t", "w_vh", "w_vh_t", "b_vh", "w_cat", "w_cat_t", "b_ih", "b_hh", "w_a", "w_a_t",
                                   "b_a", "w_m", "w_m_t", "b_m", "w_c", "b_c", "pe")] + [("f32_mask", i32), ("pad_", i32)]


class MonitorStep(C.Structure):
    _fields_ = ([(n, ptr) for n in ("prev_rep", "cand_rep", "h0", "c0", "ctx", "ctx_mask", "cand_mask", "logit", "prog", "h1", "c1",
                                    "word_w", "move_w", "pctx", "tq", "vq", "xcat", "tcat", "aq", "hm", "mg", "mem", "act", "tanh_c1",
                                    "gates", "dots", "ws")]
                + [("ws_floats", i64), ("seed_pe", u64), ("off_pe", u64), ("p_pe", f32), ("seed", u64), ("off_h1", u64),
                   ("off_mem", u64), ("p_drop", f32), ("offset_base_dev", ptr)])


class DctxTerm(C.Structure):
    _fields_ = [("alpha", ptr), ("dl", ptr), ("g", ptr), ("q", ptr), ("ldg", i64), ("ldq", i64), ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("p", f32), ("pad_", f32)]


class MonitorGrads(C.Structure):
    _fields_ = ([(n, ptr) for n in ("dlogit", "dprog", "dh1", "dc1", "dww_ext", "dmw_ext", "dprev_rep", "dcand_rep", "dh0", "dc0", "dctx")]
                + [("dctx_accumulate", i32)]
                + [(n, ptr) for n in ("g_tin", "g_vh", "g_bvh", "g_ih", "g_hh", "g_bih", "g_bhh", "g_a", "g_ba", "g_m", "g_bm", "g_wc", "g_bc")]
                + [("acc", i32 * 13), ("precision", i32), ("scratch", ptr), ("scratch_floats", i64), ("defer", C.POINTER(ParamJobs)),
                   ("dctx_term", C.POINTER(DctxTerm))])


class FollowerDims(C.Structure):
    _fields_ = [(n, i32) for n in ("B", "L", "V", "C", "H", "F", "A", "D", "wtype")]


class FollowerWeights(C.Structure):
    _fields_ = [(n, ptr) for n in ("w_h", "w_h_t", "b_h", "w_v", "b_v", "w_cat", "w_cat_t", "b_ih", "b_hh", "w_tin", "w_tin_t", "w_tout",
                                   "w_tout_t", "w_act", "b_act", "w_hid", "w_hid_t", "b_hid", "w_out", "b_out", "w_v_t")]


class FollowerStep(C.Structure):
    _fields_ = ([(n, ptr) for n in ("img", "a_prev", "cands", "h0", "c0", "ctx", "ctx_mask", "logit", "h1", "c1", "word_w", "view_w",
                                    "tq", "keys", "vlog", "xcat", "act", "tanh_c1", "tq2", "tcat", "grounded", "target", "q", "context",
                                    "gates", "dots", "ws")]
                + [("ws_floats", i64), ("seed", u64), ("off", u64), ("p_drop", f32), ("offset_base_dev", ptr), ("attn_sync", ptr),
                   ("attn_sync_bytes", i64), ("context_ready", i32)])


class FollowerGrads(C.Structure):
    _fields_ = ([(n, ptr) for n in ("dlogit", "dh1", "dc1", "dww_ext", "dvw_ext", "da_prev", "dh0", "dc0", "dctx")]
                + [("dctx_accumulate", i32)]
                + [(n, ptr) for n in ("g_wh", "g_bh", "g_wv", "g_bv", "g_ih", "g_hh", "g_bih", "g_bhh", "g_tin", "g_tout", "g_wact", "g_bact",
                                      "g_whid", "g_bhid", "g_wout", "g_bout")]
                + [("acc", i32 * 16), ("precision", i32), ("scratch", ptr), ("scratch_floats", i64), ("defer", C.POINTER(ParamJobs)),
                   ("dctx_term", C.POINTER(DctxTerm))])


BN_MLP_MAX_LAYERS = 4                 # VLN_BN_MLP_MAX_LAYERS


class BnAffine(C.Structure):
    _fields_ = [(n, ptr) for n in ("gamma", "beta", "run_mean", "run_var", "nbt")]


class BnMlpLayer(C.Structure):
    _fields_ = ([(n, ptr) for n in ("w", "w_t", "w_f32", "b")] + [("bn", BnAffine), ("out", i32), ("pad_", i32), ("p_drop", f32),
                ("padf_", f32), ("seed", u64), ("offset", u64), ("offset2", u64)])


class BnMlp(C.Structure):
    _fields_ = [("R", i32), ("D0", i32), ("nl", i32), ("wtype", i32), ("training", i32), ("R1", i32), ("eps", f32), ("momentum", f32),
                ("bn0", BnAffine), ("layer", BnMlpLayer * BN_MLP_MAX_LAYERS), ("row_zero", ptr), ("offset_base_dev", ptr),
                ("x2", ptr), ("ldx2", i64)]


class BnMlpGradLayer(C.Structure):
    _fields_ = [("g_w", ptr), ("g_b", ptr), ("g_gamma", ptr), ("g_beta", ptr), ("acc_w", i32), ("acc_b", i32), ("acc_bn", i32), ("pad_", i32)]


class BnMlpGrads(C.Structure):
    _fields_ = [("g_gamma0", ptr), ("g_beta0", ptr), ("acc0", i32), ("pad0_", i32), ("layer", BnMlpGradLayer * BN_MLP_MAX_LAYERS),
                ("precision", i32), ("bn0_from_wgrad", i32), ("scratch", ptr), ("scratch_floats", i64), ("defer", C.POINTER(ParamJobs))]


# C struct name -> ctypes mirror: load() compares sizeof on both sides (vln_struct_size)
STRUCT_MIRRORS = {
    "vln_tick_item": TickItem, "vln_wgrad_job": WgradJob, "vln_colsum_job": ColsumJob, "vln_param_jobs": ParamJobs,
    "vln_shadow_job": ShadowJob, "vln_wsum_step": WsumStep, "vln_dot_step": DotStep, "vln_ce_step": CeStep, "vln_cat_step": CatStep, "vln_select_step": SelectStep,
    "vln_monitor_loss_step": MonitorLossStep,
    "vln_monitor_dims": MonitorDims, "vln_monitor_weights": MonitorWeights, "vln_monitor_step": MonitorStep, "vln_monitor_grads": MonitorGrads,
    "vln_follower_dims": FollowerDims, "vln_follower_weights": FollowerWeights, "vln_follower_step": FollowerStep,
    "vln_follower_grads": FollowerGrads, "vln_bn_affine": BnAffine, "vln_bn_mlp_layer": BnMlpLayer, "vln_bn_mlp": BnMlp,
    "vln_bn_mlp_grad_layer": BnMlpGradLayer, "vln_bn_mlp_grads": BnMlpGrads, "vln_gather_rollout_step": GatherRolloutStep,
    "vln_gather_ride": GatherRide, "vln_envdrop_dims": EnvDropDims, "vln_envdrop_weights": EnvDropWeights, "vln_envdrop_step": EnvDropStep, "vln_dctx_term": DctxTerm,
    "vln_envdrop_grads": EnvDropGrads,
}

# symbol -> (restype, argtypes); must list EVERY function declared in include/vln_hip.h
SIGNATURES = {
    "vln_abi_version": (i32, []),
    "vln_struct_size": (i64, [C.c_char_p]),
    "vln_last_error_string": (C.c_char_p, []),
    "vln_set_graphs": (i32, [i32]),
    "vln_set_tunable": (i32, [i32, i32]),
    "vln_debug_trivial_chain": (i32, [ptr, ptr, i32, i32, i32, ptr]),
    "vln_debug_occupy": (i32, [i32, i32, i32, ptr]),
    "vln_prof_enable": (i32, [i32, i32]),
    "vln_prof_kernel_name": (C.c_char_p, [i32]),
    "vln_prof_read": (i32, [i32, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vln_linear_fwd": (i32, [ptr, i64, ptr, i32, i64, ptr, i64, i32, i32, i32, ptr, i32, ptr, i64, ptr]),
    "vln_linear_wgrad": (i32, [ptr, i64, ptr, i64, ptr, i64, i32, i32, i32, i32, ptr, i64, ptr]),
    "vln_linear_wgrad_p": (i32, [ptr, i64, ptr, i64, ptr, i64, i32, i32, i32, i32, i32, ptr, i64, ptr]),
    "vln_wgrad_grouped": (i32, [ptr, i32, i32, i32, ptr, i64, ptr]),
    "vln_colsum": (i32, [ptr, i64, ptr, i32, i32, i32, ptr, i64, ptr]),
    "vln_colsum_grouped": (i32, [ptr, i32, i32, ptr, i64, ptr]),
    "vln_linear_fwd_slabs": (i32, [ptr, i64, ptr, i32, i64, i32, i32, i32, ptr, i64, C.POINTER(i32), ptr]),
    "vln_wgrad_ride_post": (i32, [ptr, i32, ptr, i32, i32, i32, ptr, i64, ptr]),
    "vln_wgrad_ride_flush": (i32, [ptr]),
    "vln_wgrad_ride_drop": (i32, [ptr]),
    "vln_wgrad_ride_add": (i32, [ptr, i32, i32, ptr]),
    "vln_wgrad_ride_stats": (i32, [C.POINTER(i64)]),
    "vln_transpose_cast": (i32, [ptr, i64, ptr, i32, i64, i32, i32, ptr]),
    "vln_cast_copy": (i32, [ptr, i64, ptr, i32, i64, i32, i32, ptr]),
    "vln_attn_dot": (i32, [ptr, i32, ptr, i64, ptr, i32, i32, i32, ptr]),
    "vln_attn_softmax_wsum": (i32, [ptr, i32, ptr, ptr, ptr, ptr, i64, i32, i32, i32, ptr]),
    "vln_attn_dot_multi": (i32, [C.POINTER(DotStep), i32, i32, i32, i32, i64, ptr]),
    "vln_rows_wsum_multi": (i32, [C.POINTER(WsumStep), i32, i32, i32, i32, i64, f32, ptr, i64, ptr]),
    "vln_select_rows_multi": (i32, [C.POINTER(SelectStep), i32, i32, i32, ptr]),
    "vln_rows_wsum": (i32, [ptr, i32, ptr, ptr, i64, i32, i32, i32, ptr]),
    "vln_attn_bwd": (i32, [ptr, i32, ptr, ptr, ptr, ptr, i64, ptr, i64, ptr, i64, ptr, ptr, i32, i32, i32, ptr]),
    "vln_attn_fwd_rows": (i32, [ptr, i32, ptr, i64, ptr, ptr, ptr, i64, ptr, i32, i32, i32, ptr, i64, ptr]),
    "vln_attn_bwd_rows": (i32, [ptr, i32, ptr, ptr, i64, ptr, ptr, i64, ptr, ptr, i32, i32, i32, ptr, i64, ptr]),
    "vln_attn_dctx_deferred": (i32, [ptr, ptr, ptr, i64, ptr, i64, i32, ptr, i32, i32, i32, i32, ptr, ptr]),
    "vln_lstm_pointwise_fwd": (i32, [ptr, i32, i64, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, u64, u64, f32, i32, i32, ptr]),
    "vln_lstm_pointwise_bwd": (i32, [ptr, ptr, ptr, u64, u64, f32, ptr, ptr, ptr, ptr, ptr, i32, i32, ptr]),
    "vln_dropout_mask": (i32, [ptr, i64, u64, u64, f32, ptr]),
    "vln_scale_dropout": (i32, [ptr, i64, ptr, i64, i32, i32, u64, u64, f32, ptr, ptr]),
    "vln_feat_dropout_inplace": (i32, [ptr, i32, i64, i32, i32, u64, u64, f32, ptr, ptr, ptr]),
    "vln_rmsprop_partial_floats": (i64, [ptr, i32]),
    "vln_rmsprop_clip_step": (i32, [ptr, ptr, ptr, ptr, i32, ptr, ptr, f32, f32, f32, ptr, f32, ptr]),
    "vln_adam_clip_step": (i32, [ptr, ptr, ptr, ptr, ptr, i32, ptr, ptr, f32, f32, f32, f32, i64, ptr, ptr, f32, ptr]),
    "vln_sgd_clip_step": (i32, [ptr, ptr, ptr, i32, ptr, ptr, f32, ptr, f32, ptr]),
    "vln_masked_ce_fwd": (i32, [ptr, i64, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i64, i32, ptr]),
    "vln_masked_ce_bwd": (i32, [ptr, ptr, ptr, i64, ptr, i32, i32, i64, ptr]),
    "vln_masked_ce_mean_fwd": (i32, [ptr, i64, ptr, ptr, ptr, ptr, i32, i32, i64, ptr, ptr]),
    "vln_masked_ce_mean_bwd": (i32, [ptr, ptr, ptr, ptr, ptr, i32, i32, i64, ptr]),
    "vln_masked_ce_multi_fwd": (i32, [C.POINTER(CeStep), i32, i32, i64, f32, ptr, ptr, i32, ptr, ptr]),
    "vln_masked_ce_multi_bwd": (i32, [C.POINTER(CeStep), i32, i32, i64, f32, ptr, i64, ptr, ptr]),
    "vln_attn_dctx_deferred_drop": (i32, [ptr, ptr, ptr, i64, ptr, i64, i32, ptr, i32, i32, i32, i32, ptr, ptr, ptr, ptr, ptr]),
    "vln_pe_dropout": (i32, [ptr, ptr, ptr, i32, i32, i32, u64, u64, f32, ptr]),
    "vln_monitor_head_fwd": (i32, [ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, u64, u64, f32, ptr]),
    "vln_monitor_head_bwd": (i32, [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, u64, u64, f32, ptr]),
    "vln_ew": (i32, [i32, ptr, i64, ptr, i64, i32, ptr, i64, i32, i32, ptr]),
    "vln_add_n": (i32, [ptr, i64, i32, i32, ptr, i64, ptr, i64, ptr, i64, ptr, i64, i32, ptr]),
    "vln_monitor_loss_fwd": (i32, [ptr, i64, ptr, ptr, ptr, i64, ptr, ptr, ptr, i32, f32, i32, ptr, ptr, ptr, ptr, i32, i32, i64, ptr]),
    "vln_monitor_loss_bwd": (i32, [ptr, ptr, ptr, i64, ptr, ptr, ptr, i64, i32, f32, i32, ptr, ptr, i32, i32, i64, ptr]),
    "vln_monitor_loss_multi_fwd": (i32, [C.POINTER(MonitorLossStep), i32, i32, i32, f32, i64, ptr, ptr, i32, ptr]),
    "vln_monitor_loss_multi_bwd": (i32, [C.POINTER(MonitorLossStep), i32, i32, i32, f32, i64, ptr, ptr, ptr]),
    "vln_categorical_fwd": (i32, [ptr, i64, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, u64, u64, ptr, ptr]),
    "vln_categorical_bwd": (i32, [ptr, ptr, ptr, ptr, ptr, i32, i32, ptr]),
    "vln_categorical_multi_bwd": (i32, [C.POINTER(CatStep), i32, i32, ptr, ptr, ptr]),
    "vln_bn_fwd": (i32, [ptr, i64, ptr, i64, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, f32, f32, i32, i32, u64, u64, f32, ptr, ptr, i64, ptr]),
    "vln_bn_bwd": (i32, [ptr, i64, ptr, i64, ptr, i64, ptr, ptr, ptr, ptr, i64, ptr, ptr, i32, i32, f32, i32, i32, i32, u64, u64, f32,
                         ptr, ptr, i64, ptr]),
    "vln_bn_mlp_saved_floats": (i64, [C.POINTER(BnMlp)]),
    "vln_bn_mlp_out_offset": (i64, [C.POINTER(BnMlp)]),
    "vln_bn_mlp_ws_floats": (i64, [C.POINTER(BnMlp)]),
    "vln_bn_mlp_bwd_scratch_floats": (i64, [C.POINTER(BnMlp)]),
    "vln_bn_mlp_fwd": (i32, [C.POINTER(BnMlp), ptr, i64, ptr, ptr, i64, ptr]),
    "vln_bn_mlp_bwd": (i32, [C.POINTER(BnMlp), ptr, i64, ptr, ptr, i64, ptr, i64, C.POINTER(BnMlpGrads), ptr, i64, ptr]),
    "vln_a2c_loss_fwd": (i32, [ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, f32, f32, ptr, ptr, ptr, ptr, ptr, ptr]),
    "vln_a2c_loss_bwd": (i32, [ptr, i64, ptr, ptr, ptr, i32, i32, ptr, ptr, ptr, ptr]),
    "vln_wgrad_grouped_seg": (i32, [C.POINTER(WgradJob), C.POINTER(i64), C.POINTER(i64), i32, i32, i32, i32, ptr, i64, ptr]),
    "vln_colsum_grouped_seg": (i32, [C.POINTER(ColsumJob), C.POINTER(i64), i32, i32, i32, ptr, i64, ptr]),
    "vln_wgrad_grouped_ws_floats": (i64, [C.POINTER(WgradJob), i32, i32]),
    "vln_feature_table_extent": (i32, [ptr, i64, i32]),
    "vln_debug_raise_sticky": (i32, [i32]),
    "vln_gather_rollout": (i32, [ptr, i32, ptr, C.POINTER(GatherRolloutStep), i32, i32, i32, i32, i32, i32, u64, f32, ptr, ptr]),
    "vln_gather_pano": (i32, [ptr, i32, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32, u64, u64, f32, ptr]),
    "vln_gather_cands": (i32, [ptr, i32, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32, u64, u64, f32, ptr]),
    "vln_gather_step": (i32, [ptr, i32, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32, i32, u64, u64, u64,
                              f32, ptr]),
    "vln_embed_fwd": (i32, [ptr, ptr, ptr, i32, i32, i32, u64, u64, f32, ptr, ptr]),
    "vln_embed_bwd": (i32, [ptr, ptr, ptr, ptr, i32, i32, i32, i64, u64, u64, f32, ptr, ptr]),
    "vln_embed_bwd_det": (i32, [ptr, ptr, ptr, ptr, i32, i32, i32, i32, i64, u64, u64, f32, ptr, ptr]),
    "vln_tm_to_bm": (i32, [ptr, ptr, ptr, i32, i32, i32, u64, u64, f32, ptr, ptr]),
    "vln_bm_to_tm": (i32, [ptr, ptr, i32, i32, i32, u64, u64, f32, ptr, ptr]),
    "vln_graph_stats": (i32, [C.POINTER(C.c_int64)]),
    "vln_shadow_refresh": (i32, [ptr, i32, ptr]),
    "vln_lstm_sync_ws_bytes": (i64, [i32, i32, i32]),
    "vln_lstm_sync_ws_forget": (i32, [ptr]),
    "vln_lstm_sync_seq_offset": (i64, [i32, i32, i32]),
    "vln_lstm_sync_granule_range": (i32, [i32, i32, i32, C.POINTER(i64), C.POINTER(i64)]),
    "vln_lstm_seq_fwd": (i32, [ptr, ptr, i32, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32, ptr, ptr, ptr, i64, i64, ptr, ptr]),
    "vln_lstm_seq_bwd": (i32, [ptr, ptr, i32, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32, ptr, i64, i64, ptr, ptr]),
    "vln_lstm_inproj_ok": (i32, [i32, i32, i32, i32, i32, ptr, i64]),
    "vln_lstm_wgrad_inlaunch_ok": (i32, [i32, i32, i32, i32, i32, i32, i32, ptr, i64]),
    "vln_lstm_wgrad_part_floats": (i64, [i32, i32, i32, i32]),
    "vln_lstm_seq_bwd_w": (i32, [ptr, ptr, i32, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32, ptr, i64, i64, ptr,
                                 ptr, i32, ptr, ptr, i64, ptr]),
    "vln_lstm_wgrad_reduce": (i32, [ptr, i32, i32, i32, i32, ptr, ptr, ptr, ptr, ptr]),
    "vln_lstm_seq_fwd_x": (i32, [ptr, i32, ptr, ptr, ptr, i32, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32, ptr, ptr, ptr, i64, i64,
                                 ptr, ptr]),
    "vln_tick": (i32, [C.POINTER(TickItem), i32, ptr]),
    "vln_set_persistent": (i32, [i32]),
    "vln_persistent_check": (i32, []),
    "vln_host_device_pointer": (i32, [ptr, C.POINTER(C.c_void_p)]),
    "vln_host_fetch": (i32, [ptr, i32, ptr, ptr, ptr, i64, ptr]),
    "vln_prologue": (i32, [ptr, i32, ptr, ptr, ptr, i64, ptr, i32, ptr, i32, ptr]),
    "vln_set_split_attention": (i32, [i32]),
    "vln_get_split_attention": (i32, []),
    "vln_follower_bwd_scratch_floats": (i64, [ptr]),
    "vln_follower_step_fwd": (i32, [ptr, ptr, ptr, ptr]),
    "vln_follower_step_bwd": (i32, [ptr, ptr, ptr, ptr, ptr]),
    "vln_monitor_bwd_scratch_floats": (i64, [ptr]),
    "vln_monitor_ws_floats": (i64, [ptr]),
    "vln_linear_fwd_post": (i32, [ptr, i64, ptr, i32, i64, ptr, i64, i32, i32, i32]),
    "vln_linear_fwd_post_flush": (i32, [ptr, i64, ptr]),
    "vln_layout_post": (i32, [i32, ptr, ptr, ptr, i32, i32, i32, u64, u64, f32, ptr]),
    "vln_layout_post_flush": (i32, [ptr]),
    "vln_posted_drop": (i32, []),
    "vln_colsum_post": (i32, [ptr, i32, i32]),
    "vln_colsum_post_flush": (i32, [ptr, i64, ptr]),
    "vln_lstm_handoff_stats": (i32, [ptr, ptr, ptr]),
    "vln_lstm_fwd_handoff_stats": (i32, [ptr, ptr, ptr]),
    "vln_bn0_grads_from_wgrad": (i32, [ptr, ptr, ptr, i64, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32, i32, ptr, i64, ptr]),
    "vln_gemm_rows_tiling": (i32, [i32, i32, i32, ptr, ptr, ptr]),
    "vln_monitor_step_fwd": (i32, [ptr, ptr, ptr, ptr]),
    "vln_monitor_step_bwd": (i32, [ptr, ptr, ptr, ptr, ptr]),
    "vln_envdrop_ws_floats": (i64, [C.POINTER(EnvDropDims)]),
    "vln_attn_sync_bytes": (i64, [i32]),
    "vln_attn_textk_ok": (i32, [i32, i32, i32, i32, ptr, i64]),
    "vln_attn_textk_fwd": (i32, [ptr, i32, ptr, ptr, ptr, i32, i64, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, u64, u64, f32,
                                 i32, i32, i32, ptr, i64, ptr]),
    "vln_attn_textk_bwd": (i32, [ptr, i32, ptr, ptr, ptr, i32, i64, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, u64, u64, f32,
                                 i32, i32, i32, ptr, i64, ptr]),
    "vln_host_wait": (i32, [ptr, ptr, i64, ptr, ptr]),
    "vln_host_wait_fetch": (i32, [ptr, ptr, i64, ptr, ptr, i64, ptr, ptr]),
    "vln_store_to_host": (i32, [ptr, ptr, i64, ptr]),
    "vln_envdrop_flush": (i32, [ptr]),
    "vln_envdrop_drop_pending": (i32, [ptr]),
    "vln_envdrop_step_fwd": (i32, [C.POINTER(EnvDropDims), C.POINTER(EnvDropWeights), C.POINTER(EnvDropStep), ptr]),
    "vln_envdrop_step_bwd": (i32, [C.POINTER(EnvDropDims), C.POINTER(EnvDropWeights), C.POINTER(EnvDropStep),
                                   C.POINTER(EnvDropGrads), ptr]),
}

# The ABI this binding was written against (csrc/api.hip::vln_abi_version).  Entry points change their argument lists
# between versions under the SAME names, so a stale libvln_hip.so must be refused, not called with shifted arguments.
EXPECTED_ABI = 19
SHADOW_MAX_JOBS = 24          # include/vln_hip.h VLN_SHADOW_MAX_JOBS

_lib = None
try:                                   # resolved once: the two C entry points behind torch.cuda.current_stream()
    import torch as _torch
    _raw_stream, _get_device = _torch._C._cuda_getCurrentRawStream, _torch._C._cuda_getDevice
except (ImportError, AttributeError):  # CPU-only torch builds: the modules fail loudly before they get here
    _raw_stream = _get_device = None


def load():
    """Load (once) and type the library; raises VlnError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VlnError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python __graft_entry__.py` "
            "(or `make -C curriculum-learning-for-vln_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise VlnError(f"libvln_hip.so lacks symbol {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.vln_abi_version()
    if got != EXPECTED_ABI:
        raise VlnError(f"{LIB_PATH} has ABI version {got}, this package binds version {EXPECTED_ABI}: the library is stale "
                       "(or half-built); rebuild it with `python __graft_entry__.py`")
    for cname, mirror in STRUCT_MIRRORS.items():
        want = lib.vln_struct_size(cname.encode())
        if want != C.sizeof(mirror):
            raise VlnError(f"struct {cname}: the library has {want} bytes, the ctypes mirror {mirror.__name__} has {C.sizeof(mirror)}: "
                           "_lib.py and include/vln_hip.h disagree (stale library or a binding bug)")
    _lib = lib
    return lib


def raw_stream() -> int:
    """hipStream_t of torch's current stream on the current device, without building a torch.cuda.Stream object
    (the decoder path asks ~60 times per iteration; the object route costs ~3 us each)."""
    return _raw_stream(_get_device())


def check(status: int, what: str = ""):
    if status != 0:
        msg = load().vln_last_error_string()
        raise VlnError(f"{what} failed (status {status}): {msg.decode() if msg else '?'}")
