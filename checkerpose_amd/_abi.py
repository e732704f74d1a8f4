"""ctypes binding of libcheckerpose_hip.so (C ABI: include/checkerpose_hip.h).

There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CHECKERPOSE_AMD_LIB") or os.path.join(_HERE, "libcheckerpose_hip.so")   # override: kernel A/B builds

CP_F32, CP_BF16, CP_F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
LOSS_BCE, LOSS_L1 = 0, 1


class CpConvDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("out_f32", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("Cin", C.c_int32), ("in_cstride", C.c_int32), ("in_coff", C.c_int32),
                ("R", C.c_int32), ("S", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32), ("act", C.c_int32), ("slope", C.c_float), ("ksplit", C.c_int32),
                ("o_base", C.c_int64), ("o_sb", C.c_int64), ("o_sy", C.c_int64), ("o_sx", C.c_int64),
                ("o_sc", C.c_int64)]


_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_longlong


class CpChainTail(C.Structure):       # the stride-2 fuse-layer convs computed in the 64x64 chain's tail (cp_hr_branch_chain_tail)
    _fields_ = [("packed_w", C.c_void_p), ("shift", C.c_void_p), ("out", C.c_void_p * 3), ("nconv", C.c_int32),
                ("Cout", C.c_int32 * 3), ("out_cphys", C.c_int32 * 3), ("relu", C.c_int32 * 3)]


class CpChainTailConv(C.Structure):   # a first-level fuse conv in the tail of a 36 / 72 / 144-channel chain launch (cp_hr_branch_chain_tails)
    _fields_ = [("packed_w", C.c_void_p), ("shift", C.c_void_p), ("out", C.c_void_p),
                ("kind", C.c_int32), ("Cout", C.c_int32), ("out_cphys", C.c_int32), ("relu", C.c_int32)]


class CpI2fGather(C.Structure):       # Index2Feat's gather done by cp_mlp_pair_fused_gather's loader
    _fields_ = [("patches", C.c_void_p), ("x_id", C.c_void_p), ("y_id", C.c_void_p), ("mask", C.c_void_p), ("zeros", C.c_void_p),
                ("p_cstride", C.c_int32), ("p_coff", C.c_int32), ("Hp", C.c_int32), ("Wp", C.c_int32), ("k", C.c_int32)]


class CpFuseConv(C.Structure):
    _fields_ = [("packed_w", C.c_void_p), ("affine", C.c_void_p), ("out", C.c_void_p),
                ("kind", C.c_int32), ("Cout", C.c_int32), ("out_cphys", C.c_int32), ("relu", C.c_int32)]


class CpPackItem(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32 * 9), ("src", C.c_void_p), ("dst", C.c_void_p), ("row_map", C.c_void_p),
                ("total", C.c_uint64)]


class CpWgradReduceItem(C.Structure):
    _fields_ = [("ws", C.c_void_p), ("dw", C.c_void_p), ("S", C.c_int32), ("GY", C.c_int32), ("co_blocks", C.c_int32),
                ("ci_blocks", C.c_int32), ("R", C.c_int32), ("Ssz", C.c_int32), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("taps_in_block", C.c_int32), ("dw_base", C.c_int64), ("dw_sco", C.c_int64), ("dw_sci", C.c_int64),
                ("dw_sr", C.c_int64), ("dw_ss", C.c_int64)]


CP_BN_ITEM_STATS, CP_BN_ITEM_APPLY, CP_BN_ITEM_BWD_SUMS, CP_BN_ITEM_BWD_APPLY = 0, 1, 2, 3       # include/checkerpose_hip.h
BN_GROUP_MAX = 16                      # CP_BN_GROUP_MAX


class CpBnItem(C.Structure):          # one layer's pass in a grouped BatchNorm launch (cp_bn_item_* fill it; params is opaque)
    _fields_ = [("kind", C.c_int32), ("dtype", C.c_int32), ("blocks", C.c_uint32), ("lds_bytes", C.c_uint32),
                ("params", C.c_uint64 * 24)]


CP_WGRAD_ITEM_3X3, CP_WGRAD_ITEM_3X3_SMALL, CP_WGRAD_ITEM_GENERIC_BF16, CP_WGRAD_ITEM_GENERIC_F32 = 0, 1, 2, 3
CP_WGRAD_ITEM_3X3_S2_SMALL = 4


class CpWgradItem(C.Structure):       # one layer's partial-sum launch in a grouped weight-gradient launch (params is opaque)
    _fields_ = [("kind", C.c_int32), ("blocks", C.c_uint32), ("gx", C.c_uint32), ("gy", C.c_uint32), ("params", C.c_uint64 * 20)]


class CpOptItem(C.Structure):         # one parameter tensor of a cp_adam_multi / cp_sgd_multi launch
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_uint64), ("step0", C.c_uint32),
                ("pad", C.c_uint32)]


class CpFuseBwdItem(C.Structure):
    _fields_ = [("dout", C.c_void_p), ("out", C.c_void_p), ("dsrc", C.c_void_p), ("Hs", C.c_int32), ("Ws", C.c_int32), ("CG", C.c_int32),
                ("shift", C.c_int32), ("relu", C.c_int32), ("accumulate", C.c_int32), ("total", C.c_uint64)]


class CpConvGroupItem(C.Structure):   # one layer of a grouped small-Cout conv launch (cp_conv3x3_halo_item fills it; params is opaque)
    _fields_ = [("NT", C.c_int32), ("blocks", C.c_uint32), ("lds_bytes", C.c_uint32), ("pad", C.c_uint32), ("params", C.c_uint64 * 25)]


class CpWgradDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Ho", C.c_int32),
                ("Wo", C.c_int32), ("Cout", C.c_int32), ("dy_cstride", C.c_int32), ("dy_coff", C.c_int32),
                ("Cin", C.c_int32), ("x_cstride", C.c_int32), ("x_coff", C.c_int32),
                ("R", C.c_int32), ("S", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("dw_base", C.c_int64), ("dw_sco", C.c_int64), ("dw_sci", C.c_int64), ("dw_sr", C.c_int64),
                ("dw_ss", C.c_int64)]


# name -> (restype, argtypes); exactly the symbols declared in include/checkerpose_hip.h
SIGNATURES = {
    "cp_version": (_I, []),
    "cp_device_status": (_I, [_P, _I]),
    "cp_set_deterministic": (None, [_I]),
    "cp_get_deterministic": (_I, []),
    "cp_strerror": (C.c_char_p, [_I]),
    "cp_last_kernel": (C.c_char_p, []),
    "cp_kernel_log_begin": (None, []),
    "cp_kernel_log": (C.c_char_p, []),
    "cp_chan_align": (_I, [_I]),
    "cp_packed_weight_bytes": (C.c_size_t, [_I, _I, _I, _I, _I]),
    "cp_pack_conv_weight": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "cp_conv2d_igemm": (_I, [_P, C.POINTER(CpConvDesc), _P, _P, _P, _P, _P, _P]),
    "cp_conv2d_igemm_splitk": (_I, [_I, C.c_longlong, _I, _I]),
    "cp_packed_halo_weight_bytes": (C.c_size_t, [_I, _I, _I]),
    "cp_pack_conv3x3_halo_weight": (_I, [_P, _I, _P, _I, _I, _I, _P]),
    "cp_conv3x3_halo": (_I, [_P, C.POINTER(CpConvDesc), _P, _P, _P, _P, _P, _P]),
    "cp_conv3x3_halo_group_supported": (_I, [_I, _I, _I, _I]),
    "cp_conv2x2_halo_supported": (_I, [_I, _I, _I, _I]),
    "cp_packed_conv2x2_halo_weight_bytes": (C.c_size_t, [_I, _I, _I]),
    "cp_pack_conv2x2_halo_weight": (_I, [_P, _I, _P, _I, _I, _I, _P]),
    "cp_conv2x2_halo": (_I, [_P, C.POINTER(CpConvDesc), _P, _P, _P, _P, _P, _P]),
    "cp_conv3x3_halo_item": (_I, [C.POINTER(CpConvDesc), _P, _P, _P, _P, _P, _P, C.POINTER(CpConvGroupItem)]),
    "cp_conv3x3_halo_group": (_I, [_P, _I, _P, _P, _I, C.c_uint32, C.c_uint32]),
    "cp_conv3x3_s2_small_supported": (_I, [_I, _I, _I, _I]),
    "cp_conv3x3_s2_small_weight_bytes": (C.c_size_t, [_I, _I]),
    "cp_pack_conv3x3_s2_small_weight": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "cp_conv3x3_s2_small": (_I, [_P, C.POINTER(CpConvDesc), _P, _P, _P, _P, _P]),
    "cp_conv3x3_halo_up2x_supported": (_I, [_I, _I]),
    "cp_conv3x3_halo_up2x": (_I, [_P, C.POINTER(CpConvDesc), _P, _P, _P, _P, _P]),
    "cp_conv3x3_halo_seg_supported": (_I, [_I, _I, _I]),
    "cp_conv3x3_halo_seg": (_I, [_P, C.POINTER(CpConvDesc), _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "cp_pack_conv3x3_rows_weight": (_I, [_P, _I, _P, _I, _I, _I, _P]),
    "cp_basicblock_fused": (_I, [_P, C.POINTER(CpConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "cp_bottleneck_fused": (_I, [_P, C.POINTER(CpConvDesc)] + [_P] * 14),
    "cp_packed_gemm_weight_bytes": (C.c_size_t, [_I, _I, _I]),
    "cp_pack_gemm_weight": (_I, [_P, _I, _P, _I, _I, _I, _P]),
    "cp_gemm_rows": (_I, [_P, C.POINTER(CpConvDesc), _P, _P, _P, _P, _P, _P]),
    "cp_mlp_query_fused_supported": (_I, [_I, _I, _I, _I]),
    "cp_mlp_query_fused": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _F, _P, _P, _P, _F, _P, _P, _P, _L, _L, _L, _L]),
    "cp_mlp_query_fused_t": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _F, _P, _P, _P, _F, _P, _P, _P, _L, _L, _L, _L]),
    "cp_mlp_pair_fused_supported": (_I, [_I, _I, _I]),
    "cp_mlp_pair_fused": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _F, _P, _I, _I]),
    "cp_mlp_pair_fused_t": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _F, _P, _I, _I]),
    "cp_mlp_pair_fused_gather_supported": (_I, [_I, _I, _I]),
    "cp_mlp_pair_fused_gather": (_I, [_P, C.POINTER(CpI2fGather), _P, _I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _F, _P, _I, _I]),
    "cp_mlp_pair_fused_gather_t": (_I, [_P, _I, C.POINTER(CpI2fGather), _P, _I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _F, _P, _I, _I]),
    "cp_hr_stem_weight_bytes": (C.c_size_t, [_I]),
    "cp_pack_hr_stem_weights": (_I, [_P, _P, _P, _P, _P]),
    "cp_hr_stem": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "cp_hr_chain_supported": (_I, [_I, _I, _I]),
    "cp_hr_chain_weight_bytes": (C.c_size_t, [_I, _I, _I]),
    "cp_hr_chain_affine_floats": (_I, [_I, _I, _I]),
    "cp_pack_hr_chain_weight": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "cp_hr_branch_chain": (_I, [_P, _I, _I, _I, _I, _I, C.POINTER(_P), C.POINTER(C.c_int32), _I, _P, _P, _P]),
    "cp_hr_chain_tail_supported": (_I, [_I, _I, _I]),
    "cp_hr_chain_tail_weight_bytes": (C.c_size_t, []),
    "cp_hr_chain_tail_channels": (_I, []),
    "cp_pack_hr_chain_tail_weight": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "cp_hr_branch_chain_tail": (_I, [_P, _I, _I, _I, _I, _I, C.POINTER(_P), C.POINTER(C.c_int32), _I, _P, _P, _P, C.POINTER(CpChainTail)]),
    "cp_hr_chain_tailconv_supported": (_I, [_I, _I, _I, _I, _I]),
    "cp_hr_chain_tailconv_weight_bytes": (C.c_size_t, [_I, _I, _I, _I, _I]),
    "cp_pack_hr_chain_tailconv_weight": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cp_hr_branch_chain_tails": (_I, [_P, _I, _I, _I, _I, _I, C.POINTER(_P), C.POINTER(C.c_int32), _I, _P, _P, _P, _I, C.POINTER(CpChainTailConv)]),
    "cp_upsample2x_bilinear_ac": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I]),
    "cp_fuse_sum_act": (_I, [_P, _I, _I, C.POINTER(_P), C.POINTER(C.c_int32), _P, _I, _I, _I, _I, _I, _I, _I]),
    "cp_maxpool3x3s2": (_I, [_P, _I, _P, _P, _I, _I, _I, _I]),
    "cp_edgeconv_gather_max": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F]),
    "cp_hr_fuse_out_supported": (_I, [_I, _I, _I]),
    "cp_hr_fuse_out_weight_bytes": (C.c_size_t, [_I, _I, _I]),
    "cp_hr_fuse_out_affine_floats": (_I, [_I]),
    "cp_pack_hr_fuse_out_weight": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "cp_hr_fuse_out": (_I, [_P, _P, _I, _I, _I, _I, _I, C.POINTER(CpFuseConv)]),
    "cp_edgeconv_fused_supported": (_I, [_I, _I, _I, _I]),
    "cp_edgeconv_fused_weight_bytes": (C.c_size_t, [_I, _I]),
    "cp_pack_edgeconv_fused_weight": (_I, [_P, _P, _I, _I, _P]),
    "cp_edgeconv_fused": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F]),
    "cp_pack_edgeconv_fused_weight_t": (_I, [_P, _I, _P, _I, _I, _P]),
    "cp_edgeconv_fused_t": (_I, [_P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F]),
    "cp_edgeconv_tiled_supported": (_I, [_I, _I, _I, _I, _I]),
    "cp_edgeconv_tiled_weight_bytes": (C.c_size_t, [_I, _I]),
    "cp_edgeconv_tiled_table_bytes": (C.c_size_t, [_I, _I, _I]),
    "cp_pack_edgeconv_tiled_weight": (_I, [_P, _P, _I, _I, _P]),
    "cp_edgeconv_tiled": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F]),
    "cp_pack_edgeconv_tiled_weight_t": (_I, [_P, _I, _P, _I, _I, _P]),
    "cp_edgeconv_tiled_t": (_I, [_P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F]),
    "cp_permute_rows": (_I, [_P, _P, _P, _P, _P, _I, _I, _I]),
    "cp_permute_cols": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "cp_index2feat_conv_supported": (_I, [_I, _I, _I]),
    "cp_index2feat_conv_weight_bytes": (C.c_size_t, []),
    "cp_pack_index2feat_conv_weight": (_I, [_P, _P, _P]),
    "cp_index2feat_conv": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I]),
    "cp_index2feat_conv_t": (_I, [_P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I]),
    "cp_index2feat_gather": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I]),
    "cp_bits_decode": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I]),
    "cp_correspondences": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "cp_correspondences_bbox": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "cp_pnp_ransac_scratch_bytes": (C.c_size_t, [_I, _I]),
    "cp_pnp_ransac": (_I, [_P, _P, _L, _P, _P, _I, _P, _L, _I, _I, _F, _I, C.c_uint32, _P, _P, _P, _P]),
    "cp_edgeconv_bwd_workspace_bytes": (C.c_size_t, [_I, _I, _I]),
    "cp_edgeconv_gather_max_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F]),
    "cp_index2feat_gather_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I]),
    "cp_loss_workspace_bytes": (C.c_size_t, []),
    "cp_code_loss": (_I, [_P, _I, _P, _L, _P, _L, _P, _I, _I, _I, _P, _P, _L, _P]),
    "cp_mask_loss": (_I, [_P, _P, _L, _P, _I, _I, _I, _I, _I, _P, _P, _L, _P]),
    "cp_masked_ce_loss": (_I, [_P, _P, _L, _P, _P, _I, _I, _I, _P, _P, _L, _P]),
    "cp_conv2d_wgrad": (_I, [_P, C.POINTER(CpWgradDesc), _P, _P, _P]),
    "cp_conv2d_wgrad_ws": (_I, [_P, C.POINTER(CpWgradDesc), _P, _P, _P, _P, C.c_size_t]),
    "cp_conv2d_wgrad_scratch_bytes": (C.c_size_t, [C.POINTER(CpWgradDesc)]),
    "cp_conv2d_wgrad_deferred": (_I, [_P, C.POINTER(CpWgradDesc), _P, _P, _P, _P, C.c_size_t, C.POINTER(CpWgradReduceItem)]),
    "cp_conv2d_wgrad_plan": (_I, [C.POINTER(CpWgradDesc), _P, _P, _P, _P, C.c_size_t, C.POINTER(CpWgradReduceItem)]),
    "cp_conv2d_wgrad_item": (_I, [C.POINTER(CpWgradDesc), _P, _P, _P, _P, C.c_size_t, _I, C.POINTER(CpWgradItem), C.POINTER(CpWgradReduceItem)]),
    "cp_wgrad_group": (_I, [_P, _I, _P, _P, _I, C.c_uint32]),
    "cp_opt_item_blocks": (C.c_uint32, [C.c_uint64]),
    "cp_adam_multi": (_I, [_P, _P, _P, _I, C.c_uint32, _F, _F, _F, _F, _F, _I]),
    "cp_sgd_multi": (_I, [_P, _P, _P, _I, C.c_uint32, _F, _F, _F, _I]),
    "cp_wgrad_reduce_item_blocks": (C.c_uint32, [C.POINTER(CpWgradReduceItem)]),
    "cp_wgrad_reduce_batch": (_I, [_P, _P, _P, _I, C.c_uint32]),
    "cp_weight_dgrad": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "cp_bn_workspace_bytes": (C.c_size_t, [_I]),
    "cp_bn_train_stats": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "cp_affine_act": (_I, [_P, _I, _P, _I, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _F]),
    "cp_bn_bwd_workspace_bytes": (C.c_size_t, [_I]),
    "cp_bn_train_bwd": (_I, [_P, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _F, _P, _I, _I, _P, _I, _I,
                             _I, _P, _P, _P]),
    "cp_pack_item_conv": (_I, [_I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, C.POINTER(CpPackItem)]),
    "cp_pack_item_halo": (_I, [_I, _P, _I, _I, _I, _P, C.POINTER(CpPackItem)]),
    "cp_pack_item_gemm": (_I, [_I, _P, _I, _I, _I, _P, C.POINTER(CpPackItem)]),
    "cp_pack_item_dgrad_view": (_I, [_P, _I, _I, _I, _I, _P, C.POINTER(CpPackItem)]),
    "cp_pack_item_edge_view": (_I, [_P, _I, _I, _I, _P, C.POINTER(CpPackItem)]),
    "cp_pack_item_copy_f32": (_I, [_P, _P, _I, C.POINTER(CpPackItem)]),
    "cp_pack_batch": (_I, [_P, _I, _P, _P, _I, C.c_uint32]),
    "cp_bn_acc_doubles": (C.c_size_t, [_I]),
    "cp_bn_stats_accumulate": (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    "cp_bn_apply": (_I, [_P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _I, _I, _P, _I, _I, _I, _I, _I, _F, _P, _P]),
    "cp_bn_bwd_accumulate": (_I, [_P, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _F, _P]),
    "cp_bn_bwd_apply": (_I, [_P, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _F, _P, _I, _I, _P, _I, _I,
                             _I, _P, _P]),
    "cp_bn_item_stats": (_I, [_I, _P, _I, _I, _I, _I, _P, C.POINTER(CpBnItem)]),
    "cp_bn_item_apply": (_I, [_I, _P, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _I, _I, _P, _I, _I, _I, _I, _I, _F, _P, _P,
                              C.POINTER(CpBnItem)]),
    "cp_bn_item_bwd_sums": (_I, [_I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _F, _P, C.POINTER(CpBnItem)]),
    "cp_bn_item_bwd_apply": (_I, [_I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _F, _P, _I, _I, _P, _I, _I,
                                  _I, _P, _P, C.POINTER(CpBnItem)]),
    "cp_bn_group": (_I, [_P, _I, _I, _P, _P, _I, C.c_uint32, C.c_uint32]),
    "cp_bn_train_fused": (_I, [_P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _F, _F, _P, _I, _I, _P, _I, _I, _I, _I, _I, _F, _P, _P]),
    "cp_bn_bwd_fused": (_I, [_P, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _I, _I, _P, _I, _I,
                             _I, _P, _P]),
    "cp_edge_train_workspace_bytes": (C.c_size_t, [_I, _I]),
    "cp_edgeconv_train_fwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _I, _I, _P, _P, _P, _P, _P, _P,
                                   _I, _I, _I, _I, _I, _F]),
    "cp_edgeconv_train_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P,
                                   _I, _I, _I, _I, _I, _F]),
    "cp_edge_weight_view": (_I, [_P, _P, _I, _I, _I, _P]),
    "cp_upsample2x_bilinear_ac_bwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "cp_fuse_sum_act_bwd": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I]),
    "cp_fuse_sum_act_bwd_item": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(CpFuseBwdItem), C.POINTER(C.c_uint32)]),
    "cp_fuse_sum_act_bwd_group": (_I, [_P, _I, _P, _P, _I, C.c_uint32]),
    "cp_maxpool3x3s2_bwd": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I]),
    "cp_memset_zero": (_I, [_P, _P, C.c_size_t]),
    "cp_strided_to_nhwc": (_I, [_P, _I, _P, _I, _L, _L, _L, _L, _P, _I, _I, _I, _I]),
    "cp_memcpy_d2d": (_I, [_P, _P, _P, C.c_size_t]),
    "cp_index2feat_gather_bwd_t": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I]),
    "cp_nchw_to_nhwc": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I]),
    "cp_u8hwc_to_nhwc_norm": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "cp_crop_resize_u8": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I]),
    "cp_nhwc_to_nchw_f32": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I]),
    "cp_graph_begin_capture": (_I, [_P]),
    "cp_graph_end_capture": (_I, [_P, C.POINTER(_P)]),
    "cp_graph_capture_set_deps": (_I, [_P, C.POINTER(_P), _I]),
    "cp_graph_capture_tail": (_I, [_P, C.POINTER(_P), _I, C.POINTER(_I)]),
    "cp_graph_launch": (_I, [_P, _P]),
    "cp_graph_destroy": (_I, [_P]),
}

_lib = None


def load():
    """Load the library once; raise loudly if it was not built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        # torch ships its own libamdhip64: it must be resident BEFORE our library is dlopen()ed, otherwise the loader binds
        # us to /opt/rocm's copy and the process ends up with two HIP runtimes (every launch on a torch stream then fails)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("checkerpose_amd: HIP library %s is missing -- build it with "
                               "`make -C checkerpose_amd/csrc` (there is no CPU/PyTorch fallback)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the .so is stale: also loud
            fn.restype, fn.argtypes = res, args
        if os.environ.get("CHECKERPOSE_AMD_DETERMINISTIC", "0") == "1":     # deterministic training mode (include/checkerpose_hip.h)
            lib.cp_set_deterministic(1)
        _lib = lib
    return _lib


STATUS_BITS = {1: "CP_STATUS_CHAIN0_HANDOVER: the pipelined 64 x 64 HRNet chain (hr_chain0p_kernel) gave up a bounded wait for a row "
                  "hand-over between its waves -- the outputs of that forward are wrong",
               2: "CP_STATUS_CHAIN0_STAGING: hr_chain0p_kernel's staging / tail loop ran out of its bound before finishing its rows"}


def device_status(clear=True):
    """the current device's sticky status word (include/checkerpose_hip.h: cp_device_status); synchronises the device"""
    w = C.c_uint32(0)
    check(load().cp_device_status(C.byref(w), 1 if clear else 0), "cp_device_status")
    return int(w.value)


def raise_on_device_status(where=""):
    """RuntimeError if a kernel reported a failure of its own since the last look (the word is cleared: the error is raised once)"""
    w = device_status(clear=True)
    if w:
        msgs = [m for b, m in STATUS_BITS.items() if w & b] or ["unknown bits"]
        raise RuntimeError("checkerpose_hip: device status %#x%s -- %s" % (w, (" (" + where + ")") if where else "", "; ".join(msgs)))


def check(code, what=""):
    if code != 0:
        msg = load().cp_strerror(code).decode()
        raise RuntimeError("checkerpose_hip %s failed: %s (code %d)" % (what, msg, code))


def device_table(items, blocks, device):
    """The device-side form of a grouped launch: `items` (ctypes structs of ONE type, the kernels' parameter blocks) as a byte tensor
    on `device` plus the exclusive prefix sum (len(items) + 1 uint32 entries) of `blocks` (workgroups per item) -- block b of the launch
    belongs to item k with prefix[k] <= b < prefix[k + 1].  -> (items tensor, prefix tensor, total blocks); keep both tensors alive."""
    import torch
    arr = (type(items[0]) * len(items))(*items)
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    pre = [0]
    for nb in blocks:
        pre.append(pre[-1] + int(nb))
    prefix = torch.tensor(pre, dtype=torch.int64).to(torch.int32).to(device)
    return raw, prefix, pre[-1]
