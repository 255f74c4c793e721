"""ctypes binding of liblanemap_hip.so — the C-ABI declared in include/lanemap_hip.h.

The library is the product: there is no CPU/ATen fallback.  If it is missing, import of any op
fails loudly with a build hint (run ``python -m lanemapping_amd.build``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('LANEMAP_HIP_LIB') or os.path.join(_HERE, 'liblanemap_hip.so')   # override: kernel experiments only

c_f32p = C.POINTER(C.c_float)
vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_long, C.c_float


class LmRasterParams(C.Structure):
    """Per-tile BEV<->LAS parameters (reference utils/io_utils.py:125-150)."""
    _fields_ = [('quat', C.c_float * 4), ('trans', C.c_float * 3), ('bev_img_offset', C.c_float * 2),
                ('img_reso', C.c_float * 2), ('local_min_ele', C.c_float), ('ele_reso', C.c_float),
                ('inten_lo', C.c_float), ('inten_hi', C.c_float)]


class LmLasHeader(C.Structure):
    """Fields of the LAS public header block the ingest needs (include/lanemap_hip.h)."""
    _fields_ = [('version_major', C.c_int), ('version_minor', C.c_int), ('point_format', C.c_int), ('record_len', C.c_int),
                ('n_points', C.c_long), ('offset_to_points', C.c_long), ('scale', C.c_double * 3), ('offset', C.c_double * 3),
                ('min_xyz', C.c_double * 3), ('max_xyz', C.c_double * 3)]


# name -> (restype, argtypes); every entry must be declared in include/lanemap_hip.h
SIGNATURES = {
    'lm_abi_version': (i32, []),
    'lm_last_error': (C.c_char_p, []),
    'lm_device_count': (i32, []),
    'lm_conv2d_nhwc_mfma_f32': (i32, [vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, vp, i32,
                                      i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32]),
    'lm_conv2d_nhwc_mfma_resup_f32': (i32, [vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, vp, i32,
                                            i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32]),
    'lm_winograd44_supported': (i32, [i32, i32, i32, i32]),
    'lm_winograd44_gn_chunks': (i32, [i32, i32, i32]),
    'lm_winograd44_tiles': (i64, [i32, i32, i32, i32]),
    'lm_winograd44_twin_workspace_bytes': (i64, [i32, i32, i32, i32, i32, i32]),
    'lm_conv3x3_winograd44_f32': (i32, [vp, vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    'lm_conv3x3_winograd44_twin_f32': (i32, [vp, vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64]),
    'lm_wino44_split_fragments': (i32, [vp, vp, vp, i64]),
    'lm_conv3x3_winograd44_split_f32': (i32, [vp, vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, f32]),
    'lm_conv3x3_winograd44_split_twin_f32': (i32, [vp, vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, f32]),
    'lm_conv2d_nhwc_mfma_f32_gnstats': (i32, [vp, vp, i32, vp, i32, vp, vp, i32, vp] + [i32] * 11),
    'lm_gn_finalize': (i32, [vp, vp, vp, i32, i32, i32, i32, f32]),
    'lm_gn_finalize_split': (i32, [vp, vp, vp, i32, i32, i32, i32, f32, i32]),
    'lm_stem_conv7x7_bn_relu': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32]),
    'lm_stem_conv7x7_bn_relu_u8': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32]),
    'lm_maxpool3x3s2_nhwc': (i32, [vp, vp, vp, i32, i32, i32, i32]),
    'lm_conv2d_nhwc_small': (i32, [vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32]),
    'lm_gn_stats': (i32, [vp, vp, vp, vp, i32, i32, i32, f32]),
    'lm_gn_stats_workspace_bytes': (i64, [i32, i32, i32]),
    'lm_gn_relu_upsample': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32]),
    'lm_gn_relu_upsample_sum': (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32]),
    'lm_gn_relu_upsample_sum_conv1x1': (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, vp, i32]),
    'lm_upsample_bilinear_nhwc': (i32, [vp, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32]),
    'lm_upsample_bilinear_to_chw': (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32]),
    'lm_layernorm_rows': (i32, [vp, vp, vp, vp, vp, i64, i32, f32]),
    'lm_unpatchify': (i32, [vp, vp, vp, i32, i32, i32, i32]),
    'lm_attention_f32': (i32, [vp, vp, vp, i32, i32, i32, i32, f32]),
    'lm_attention_masked_f32': (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32]),
    'lm_head_tokens': (i32, [vp, vp, vp, vp, f32, i32, i32, i32, i32, i32, i32]),
    'lm_head_stage2': (i32, [vp, vp, i32, i32, vp, vp, vp, vp, vp, i64]),
    'lm_head_proposal_conf': (i32, [vp, vp, vp, vp, vp, i32, i32]),
    'lm_decode_proposals': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, i32]),
    'lm_decode_orient': (i32, [vp, vp, i32, i32, vp, i64]),
    'lm_decode_semantic': (i32, [vp, vp, vp, vp, vp, i32, i32, i32, f32, i32]),
    'lm_endp_topk_workspace_bytes': (i64, [i32]),
    'lm_endp_topk': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32]),
    'lm_pack_segments': (i32, [vp, i32, vp, vp, vp, vp]),
    'lm_bev_raster_workspace_bytes': (i64, [i32, i64, i32, i32]),
    'lm_bev_raster_batch': (i32, [vp, vp, C.POINTER(i64), C.POINTER(LmRasterParams), i32, vp, i64, vp, vp, i32, i32]),
    'lm_tile_ingest_u8': (i32, [vp, vp, vp, i32, i32, i32, i32]),
    'lm_endp_cluster': (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp]),
    'lm_polyline_assemble': (i32, [vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, vp]),
    'lm_raster_polylines': (i32, [vp, i32, i32, vp]),
    'lm_line8': (i32, [vp, i32, i32, i32, i32, i32]),
    'lm_trace_lines': (i32, [vp, i32, i32, vp, vp]),
    'lm_polyline_backproject': (i32, [vp, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp]),
    'lm_skeletonize_lee_2d': (i64, [vp, i32, i32]),
    'lm_merge_create': (vp, []),
    'lm_merge_destroy': (None, [vp]),
    'lm_merge_add_tile': (i32, [vp, vp, vp, i32]),
    'lm_merge_finish': (i64, [vp, C.POINTER(i64)]),
    'lm_merge_result': (i32, [vp, vp, vp]),
    'lm_downsample_seq': (i32, [vp, i32, C.c_double, vp]),
    'lm_softmax_rows': (i32, [vp, vp, i64, i32]),
    'lm_rowref_select': (i32, [vp, vp, vp, vp, vp, f32, vp, i32, i32, i32, i32]),
    'lm_rowref_gather': (i32, [vp, vp, vp, vp, i32, i32, i32, i32]),
    'lm_rowref_scatter': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32]),
    'lm_rowref_decode': (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32]),
    'lm_voxelize_workspace_bytes': (i64, [i64]),
    'lm_scan_workspace_bytes': (i64, [i64]),
    'lm_exclusive_scan_u32': (i32, [vp, vp, vp, i64, vp, i64]),
    'lm_sort_pairs_workspace_bytes': (i64, [i64]),
    'lm_sort_pairs_u32': (i32, [vp, vp, vp, i64, i32, vp, i64]),
    'lm_voxelize_hard': (i32, [vp, vp, i64, c_f32p, c_f32p, C.POINTER(i32), i32, i32, i32, vp, i32, vp, i32, vp, vp, i32, vp, i64]),
    'lm_sparse_grid_build': (i32, [vp, vp, i64, vp, i32, i32, i32, i32]),
    'lm_sparse_conv_outputs_workspace_bytes': (i64, [i64]),
    'lm_sparse_conv_outputs': (i32, [vp, vp, i64, i32, C.POINTER(i32), i32, i32, i32, vp, vp, i32, vp, vp, i64]),
    'lm_sparse_rulebook': (i32, [vp, vp, i64, vp, i32, i32, i32, i32, C.POINTER(i32), vp]),
    'lm_conv_gather_mfma_f32': (i32, [vp, vp, i32, vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, i64, i32, i32, i32]),
    'lm_sparse_to_dense_nhwc': (i32, [vp, vp, i32, vp, i64, vp, i32, i32, i32, i32, i32, i32]),
    'lm_upsample_bicubic_nhwc': (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32]),
    'lm_las_parse_header': (i32, [vp, i64, C.POINTER(LmLasHeader)]),
    'lm_las_decode_points': (i32, [vp, vp, i32, i64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), f32, f32,
                                   i32, vp]),
    'lm_png_info': (i32, [vp, i64, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    'lm_png_decode_u8': (i32, [vp, i64, vp, i64]),
    'lm_png_decode_files_u8': (i32, [C.POINTER(C.c_char_p), i32, vp, i32, i32, i32, i32]),
    'lm_zlib_inflate': (i32, [vp, i64, vp, i64, C.POINTER(i64)]),
    'lm_lane_json_text': (i64, [vp, i32, i32, i32, vp, i64]),
    'lm_lane_json_write': (i32, [vp, i32, i32, i32, C.c_char_p]),
    'lm_seqs_json_write': (i32, [vp, vp, i32, i32, i32, C.c_char_p]),
}

_lib = None


class LanemapHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LanemapHipError(
                f'{LIB_PATH} is missing: the HIP library is the only implementation of this package '
                '(no CPU fallback). Build it with `python -m lanemapping_amd.build`.')
        import torch  # noqa: F401  first: the process must hold ONE HIP runtime, the one torch loads (loading ours before
        #                     torch's made every later launch fail with "no ROCm-capable device is detected")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)           # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code != 0:
        raise LanemapHipError(f'lanemap_hip error {code}: {lib().lm_last_error().decode()}')
