/* lanemap_hip.h — C ABI of liblanemap_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the inference hot path of WHU-USI3DV/LaneMapping.  The reference is pure Python
 * (torch.nn modules built through a registry); what a maintainer binds from the reference side
 * is therefore a ctypes stub per torch.nn call site.  Every entry point:
 *   - takes plain pointers + sizes (device pointers unless the name says host), no torch types;
 *   - launches on the hipStream_t passed as `stream` (NULL = default stream) and returns immediately;
 *   - returns 0 on success, non-zero on error (lm_last_error() gives the message; nothing is thrown);
 *   - borrows its inputs (const), never frees or allocates caller-visible memory.
 * Activations are fp32, NHWC ("pixel-major rows", channel stride 1); `ld*` = floats between consecutive pixels.
 * Citations are file:line in the reference (relative to its repo root).
 */
#ifndef LANEMAP_HIP_H
#define LANEMAP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { LM_OK = 0, LM_ERR_ARG = 1, LM_ERR_HIP = 2, LM_ERR_NO_DEVICE = 3 };
enum { LM_ACT_NONE = 0, LM_ACT_RELU = 1, LM_ACT_GELU = 2 };

int lm_abi_version(void);
const char* lm_last_error(void);          /* thread-local, valid until the next failing call */
int lm_device_count(void);

/* ---- matrix-core convolution / GEMM -------------------------------------------------------------------------
 * y[m,n] = act((sum_{tap,c} x[pix(m,tap),c] * wp[tap][n][c]) * scale[n] + shift[n] + res[m % res_rows, n])
 * Replaces every nn.Conv2d with Cin%32==0 of FPNWrapper (baseline/models/pcencoder/postprojector.py:463-511,
 * 563-655; BasicBlock :299-338 with BN folded into scale/shift), every nn.Linear of VitSegNet
 * (baseline/models/backbone/vitsegnet.py:32-35,51-56,165; patch embedding = 8x8 stride-8 case) and the first
 * Conv1d+BN1d of ext2/cls2/offset2 (baseline/models/heads/polyline_fpn_vit_vertex_2.py:206-228).
 * wp: [KH*KW][CoutP][Cin], CoutP = Cout rounded up to 128 (zero rows).  scale/shift/res may be NULL.
 * res_rows == 0: residual has one row per output pixel; > 0: row index is m % res_rows (positional embedding). */
int lm_conv2d_nhwc_mfma_f32(void* stream, const float* x, int ldx, const float* wp, int CoutP,
                            const float* scale, const float* shift, const float* res, int ldr, int res_rows,
                            float* y, int ldy, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                            int stride, int pad_h, int pad_w, int dil, int act);
/* Same convolution with a COARSE residual [B][Hr][Wr][ldr] added through bilinear (align_corners=True) interpolation to Ho x Wo:
 * `_upsample_add(p_coarse, latlayer(c))` of the FPN (postprojector.py:549-561, 595-601) without writing the upsampled map.
 * Cout, ldy, ldr multiples of 4.  Bit-identical to lm_upsample_bilinear_nhwc + lm_conv2d_nhwc_mfma_f32(res = that map). */
int lm_conv2d_nhwc_mfma_resup_f32(void* stream, const float* x, int ldx, const float* wp, int CoutP, const float* scale,
                                  const float* shift, const float* res_coarse, int ldr, int Hr, int Wr, float* y, int ldy,
                                  int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_h, int pad_w,
                                  int dil, int act);

/* Winograd F(4x4,3x3) on the fp32 matrix cores (csrc/conv_wino44.hip): the 3x3 / stride-1 layers (postprojector.py:322-338,597-647) with 36
 * products per 4x4 output block instead of 144 (3x3 / stride 1 / pad == dilation), exact fp32 MFMA, no transformed tensor in HBM.
 * Error at the level of a re-ordered direct sum (transform constants up to 8 and down to 1/24: profiles/r3_f44_numerics_study.txt);
 * lm_conv3x3_winograd44_twin_f32 is the materialising twin (V and M in a workspace, three plain kernels) with identical bits.
 * wu_frag: U = G g G^T (fp64 -> fp32) per wave fragment, [36][Cin/8][CoutP/32][64 lanes][4]:
 *   wu_frag[xi][u][nt][lane][e] = U[xi][nt*32 + (lane & 31)][8 u + 4 (lane >> 5) + e], CoutP % 64 == 0, Cin % 16 == 0.
 * wu (twin): U as [36][CoutP][Cin].  gn_partial: [B][lm_winograd44_gn_chunks][Cout][2] doubles -> lm_gn_finalize. */
int lm_winograd44_supported(int H, int W, int Cin, int dil);
int lm_winograd44_gn_chunks(int H, int W, int dil);
long lm_winograd44_tiles(int B, int H, int W, int dil);
long lm_winograd44_twin_workspace_bytes(int B, int H, int W, int Cin, int CoutP, int dil);
int lm_conv3x3_winograd44_f32(void* stream, const float* x, int ldx, const float* wu_frag, int CoutP, const float* scale,
                              const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W,
                              int Cin, int Cout, int dil, int act, double* gn_partial);
int lm_conv3x3_winograd44_twin_f32(void* stream, const float* x, int ldx, const float* wu, int CoutP, const float* scale,
                                   const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W,
                                   int Cin, int Cout, int dil, int act, void* workspace, long workspace_bytes);
/* SECOND LINE of the same convolution (never the default path; round 6): the Winograd-domain products on the fp16 matrix pipe with
 * fp32-accurate products - every fp32 operand x = hi + lo in two fp16 terms (22 bits), v u ~= v_hi u_hi + v_hi u_lo + v_lo u_hi as three
 * v_mfma_f32_32x32x8_f16 with fp32 accumulation.  The caller scales U by a power of two u_scale (max |U| u_scale ~ 2^13), packs its
 * fragments as for lm_conv3x3_winograd44_f32 and splits them ONCE with lm_wino44_split_fragments (n_quads = elements / 4); post = 1 / u_scale
 * is folded into the epilogue's scale (exact).  Needs |activation| < ~650 (fp16 range of the transformed input; not checked).
 * lm_conv3x3_winograd44_split_twin_f32: the materialising twin with identical bits (wu = U * u_scale as [36][CoutP][Cin] fp32). */
int lm_wino44_split_fragments(void* stream, const float* frag, float* out, long n_quads);
int lm_conv3x3_winograd44_split_f32(void* stream, const float* x, int ldx, const float* wu_split, int CoutP, const float* scale,
                                    const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W,
                                    int Cin, int Cout, int dil, int act, double* gn_partial, float post);
int lm_conv3x3_winograd44_split_twin_f32(void* stream, const float* x, int ldx, const float* wu, int CoutP, const float* scale,
                                         const float* shift, const float* res, int ldr, float* y, int ldy, int B, int H, int W,
                                         int Cin, int Cout, int dil, int act, void* workspace, long workspace_bytes, float post);

/* Same convolution + first pass of GroupNorm(C,C) (postprojector.py:512-515,608-647): also writes per (image, 64-row
 * chunk, channel) sum / sum of squares of the outputs, gn_partial [B][Ho*Wo/64][Cout][2] doubles -> lm_gn_finalize. */
int lm_conv2d_nhwc_mfma_f32_gnstats(void* stream, const float* x, int ldx, const float* wp, int CoutP, const float* shift,
                                    float* y, int ldy, double* gn_partial, int B, int H, int W, int Cin, int Cout,
                                    int KH, int KW, int stride, int pad_h, int pad_w, int dil);
int lm_gn_finalize(void* stream, const double* partial, float* stats, int B, int HW, int C, int nchunk, float eps);
/* same values, laid out per channel group: stats [split][B][C / split][2] (two branches sharing one merged convolution) */
int lm_gn_finalize_split(void* stream, const double* partial, float* stats, int B, int HW, int C, int nchunk, float eps, int split);

/* ---- thin layers ---------------------------------------------------------------------------------------------
 * stem: relu(bn1(conv1(x))) for planar x [B,3,H,W] -> NHWC [B,H/2,W/2,64]; w_k64 = [7][7][3][64]
 * (postprojector.py:458-460,566).  maxpool: 3x3 stride 2 pad 1 (:461,567).
 * small: direct conv for Cout <= 16 (feature_layer/output_layer_* :509-511,628-651; head_common_layers,
 * orient, bi_seg_proposal heads/polyline_fpn_vit_vertex_2.py:183-189,232-237,249); w_tc16 = [KH*KW][Cin][16];
 * y = act(conv(pre_relu ? relu(x) : x) * scale + shift). */
int lm_stem_conv7x7_bn_relu(void* stream, const float* x_chw, const float* w_k64, const float* scale,
                            const float* shift, float* y_nhwc, int B, int H, int W);
/* The stem on a u8 HWC tile [B][H][W][3] as the rasteriser / PNG reader emit it (u8 / 255 = the reference's to_tensor,
 * datasets/laserlane_proposals.py:85-98, applied while staging): same bits as tile_ingest + the f32 stem, a quarter of the bytes. */
int lm_stem_conv7x7_bn_relu_u8(void* stream, const unsigned char* x_hwc3, const float* w_k64, const float* scale,
                               const float* shift, float* y_nhwc, int B, int H, int W);
int lm_maxpool3x3s2_nhwc(void* stream, const float* x, float* y, int B, int H, int W, int C);
int lm_conv2d_nhwc_small(void* stream, const float* x, int ldx, const float* w_tc16, const float* scale,
                         const float* shift, float* y, int ldy, int B, int H, int W, int Cin, int Cout,
                         int KH, int KW, int stride, int pad_h, int pad_w, int pre_relu, int act);

/* ---- normalisation / resampling (postprojector.py:512-515,541-561,608-651; vitsegnet.py:20-26,180) ----------
 * gn_stats: per-(b,c) mean, rstd of GroupNorm(C,C) -> stats [B][C][2]; workspace from lm_gn_stats_workspace_bytes.
 * gn_relu_upsample: y (= | +=) bilinear_align_corners(relu(gn(x))) to Ho x Wo.
 * upsample_bilinear_*: F.interpolate(mode='bilinear', align_corners=True); `add` (optional) is summed in. */
int lm_gn_stats(void* stream, const float* x, double* workspace, float* stats, int B, int HW, int C, float eps);
long lm_gn_stats_workspace_bytes(int B, int HW, int C);
int lm_gn_relu_upsample(void* stream, const float* x, const float* stats, const float* gamma, const float* beta,
                        float* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int accumulate);
/* y = ((t0 + t1) + t2), t_k = bilinear_align_corners(relu(gn(x[k]; stats[k], gamma, beta))) from Hi[k] x Wi[k] to Ho x Wo, n <= 3
 * terms sharing gamma / beta: `s2 + s3 + s4` of one semantic branch (postprojector.py:615-621, :641-647) in one pass. */
int lm_gn_relu_upsample_sum(void* stream, int n, const float* const* x, const float* const* stats, const int* Hi, const int* Wi,
                            const int* ldx /* floats between pixels per term, NULL = C */, const float* gamma, const float* beta,
                            float* y, int B, int Ho, int Wo, int C);
/* The same sum followed by a 1x1 convolution y1[pixel][0..cout) = sum[pixel][:] @ w + bias (cout <= 8; w_c16 = [C][16] layout of
 * lm_conv2d_nhwc_small; C/4 a power of two <= 64): feature_layer / output_layer_endp (postprojector.py:628-651).  y may be NULL: the
 * C-channel sum is then never written. */
int lm_gn_relu_upsample_sum_conv1x1(void* stream, int n, const float* const* x, const float* const* stats, const int* Hi, const int* Wi,
                                    const int* ldx, const float* gamma, const float* beta, float* y, int B, int Ho, int Wo, int C,
                                    const float* w_c16, const float* bias, int cout, float* y1, int ldy1);
int lm_upsample_bilinear_nhwc(void* stream, const float* x, int ldx, const float* add, int lda, float* y, int ldy,
                              int B, int Hi, int Wi, int Ho, int Wo, int C);
int lm_upsample_bilinear_to_chw(void* stream, const float* x, int ldx, float* y_chw, int B, int Hi, int Wi,
                                int Ho, int Wo, int C);
int lm_layernorm_rows(void* stream, const float* x, const float* gamma, const float* beta, float* y,
                      long rows, int D, float eps);
int lm_unpatchify(void* stream, const float* tokens, float* y_nhwc, int B, int G, int P, int C);

/* ---- attention core: softmax(q k^T * scale) v per (batch, head); qkv = [B*N][3*heads*64] (vitsegnet.py:58-68) */
int lm_attention_f32(void* stream, const float* qkv, float* out, int B, int N, int heads, int dim_head, float scale);
/* the same with a key mask, valid [B][N] ints (N <= 64): per batch element only the flagged tokens are keys, compacted in token order
 * (row_shared_not_reduc_ref.py:199-215: the transformer runs over the data-dependent subset of the lane tokens) */
int lm_attention_masked_f32(void* stream, const float* qkv, float* out, const int* valid, int B, int N, int heads, int dim_head, float scale);

/* ---- column-proposal head (heads/polyline_fpn_vit_vertex_2.py:390-421) ----------------------------------------
 * tokens: tok[(b,p,h), c*10+w] = avg_pool8(up(seg window p))[h,w] * row_fea_pad[b,c,h,2p+w]   (:392-405)
 * stage2: second Conv1d of ext2/cls2/offset2 (:210,218,226); proposal_conf: Linear(23040 -> 2) (:200-204) */
int lm_head_tokens(void* stream, const float* seg, const float* row_nhwc16, float* tok, float seg_bias,
                   int B, int P, int Hr, int Wr, int prop_width, int half_buff);
int lm_head_stage2(void* stream, const float* hid, int ldh, int D, const float* w2, const float* b2,
                   float* ext2, float* cls2, float* off2, long M);
int lm_head_proposal_conf(void* stream, const float* tok, const float* wt, const float* bias, float* conf,
                          int BP, int L);

/* ---- decode (heads/polyline_fpn_vit_vertex_2.py:602-759; postprojector.py:115-183) ---------------------------
 * proposals: :610, :694-697, :701-702, :726-738.  orient: :615.  semantic: :627-632 (raw_mode=1: :122-127).
 * endp_topk: sigmoid of logits cropped by `clip` px, K best in (score desc, flat index asc) order (:647-668);
 * out_status[b] = 1 if more tied scores than the candidate buffer holds. */
int lm_decode_proposals(void* stream, const float* pconf, const float* ext2, const float* cls2, const float* off2,
                        float* prop_conf, float* v_ext, float* cls_conf, int* cls_idx, double* cls_offset,
                        int B, int P, int R, float exist_thre, int prop_width, int half_buff);
int lm_decode_orient(void* stream, const float* x_nhwc, int ldx, int C, unsigned char* y, long pixels);
int lm_decode_semantic(void* stream, const float* logit_chw3, unsigned char* sem, float* biseg, float* rows,
                       int B, int H, int W, float thre, int raw_mode);
long lm_endp_topk_workspace_bytes(int B);
int lm_endp_topk(void* stream, const float* endp_logit, void* workspace, int* out_idx, float* out_score,
                 int* out_status, int B, int H, int W, int clip, int K);

/* Gathers up to 8 device segments (bytes[s] % 4 == 0) into one block at dst + dst_offsets[s] (% 16 == 0): the decode outputs the host
 * post-processing reads (prop_conf, v_ext, cls_offset, rows, idx, status) then travel in ONE device-to-host copy per batch - the per-batch
 * body of Runner.infer_lane_coordinate_endpoint_semantics (baseline/engine/runner.py:725-740) moves them with one .cpu() per tensor. */
int lm_pack_segments(void* stream, int n, const void* const* src, const long* bytes, const long* dst_offsets, void* dst);


/* ---- LAS -> BEV rasteriser and tile ingest (build-defined, parity unpinned: the reference has no rasteriser;
 * pinned pieces: datasets/laserlane_proposals.py:85-98,618-636; utils/coor_img2pc.py:127-183;
 * utils/io_utils.py:125-150) */
typedef struct {
    float quat[4];            /* [w,x,y,z] = las_rotation_trans_quan[3:7] */
    float trans[3];           /* las_rotation_trans_quan[0:3] */
    float bev_img_offset[2];
    float img_reso[2];
    float local_min_ele;
    float ele_reso;
    float inten_lo, inten_hi; /* 800, 33000 */
} LmRasterParams;
/* points: device [sum N][4] f32 {x,y,z,raw intensity}; tile_offsets: HOST [B+1] point index of each tile's first
 * record; params: HOST [B]; out_chw [B][3][H][W] f32 (= u8/255), out_hwc_u8 [B][H][W][3] (either may be NULL). */
long lm_bev_raster_workspace_bytes(int B, long max_points_per_tile, int H, int W);
int lm_bev_raster_batch(void* stream, const float* points_xyzi, const long* tile_offsets, const LmRasterParams* params,
                        int B, void* workspace, long workspace_bytes, float* out_chw, unsigned char* out_hwc_u8,
                        int H, int W);
int lm_tile_ingest_u8(void* stream, const unsigned char* src_hwc, float* dst_chw, int B, int H, int W, int C);

/* ---- host-side tail (HOST pointers; no GPU is touched) --------------------------------------------------------
 * endp_cluster: heads/polyline_fpn_vit_vertex_2.py:661-688 + :903-924.
 * polyline_assemble: :805-861 + baseline/utils/polyline_utils.py (whole file) + :1091-1115.
 * raster_polylines: polyline_utils.py:610-638 (own Bresenham instead of cv2.line). */
int lm_endp_cluster(const int* topk_idx, int n_avail, int Wc, int clip, int k0, int k_step, int k_max, int radius,
                    int min_clusters, int* out_hw, int max_out, int* n_out, int* k_used);
int lm_polyline_assemble(const float* prop_conf, const float* prop_v_ext, const double* cls_offset,
                         const float* bi_seg_rows, const int* endp_hw, int n_endp, int P, int R, float obj_thre,
                         int min_vertices, double* out_lanes, int* endp_keep);
int lm_raster_polylines(const double* lanes, int P, int R, unsigned char* out);
/* One segment of that rasteriser (cv2.line, thickness 1, LINE_8, restated from OpenCV's LineIterator: csrc/postproc.cpp) on a 1152 x 1152 map. */
int lm_line8(unsigned char* out, int x1, int y1, int x2, int y2, int colour);
int lm_trace_lines(const double* cols, int n, int R, const float* seg_rows, double* out);   /* polyline_utils.py:222-387 */
/* BEV polylines -> LAS frame: baseline/utils/coor_img2pc.py:127-183 (+ :22-53, :59-73, :94-122).  bev_hwc [H][W][C] u8 is
 * modified in place (elevation fill of empty vertex pixels, like the reference); img_seqs [L][Vmax][2] (row, col);
 * params13 = img_reso[2], bev_img_offset[2], ele_reso, local_min_ele, las_rotation_trans_quan[7]; out [L][Vmax][3]. */
int lm_polyline_backproject(unsigned char* bev_hwc, int H, int W, int C, const double* img_seqs, const int* seq_lens, int L,
                            int Vmax, const double* params13, const double* las_read_offset, double* out);

/* ---- evaluation: Lee-Kashyap-Chu thinning of a 2-D binary image, the skeletonisation inside the reference's semantic-line F1
 * (baseline/utils/metric_utils.py:415-481 -> skimage.morphology.skeletonize(method='lee')).  PARITY UNPINNED (skimage absent: the
 * published algorithm is restated, csrc/skeleton.cpp).  img [H][W] u8 (nonzero = object) is thinned in place to 0 / 1; returns the
 * number of deleted pixels, -1 on bad arguments. */
long lm_skeletonize_lee_2d(unsigned char* img, int H, int W);

/* ---- cross-tile merge of LAS-frame polylines into map-level lines (baseline/utils/merge_lines.py) ------------------------
 * Streaming host merger: create, one lm_merge_add_tile per tile in sorted file-name order (n polylines, points concatenated
 * [sum lens][3] doubles, every polyline >= 2 vertices; n = 0 for a tile without usable lines), lm_merge_finish (returns the
 * number of merged lines, *total_points their vertex count), lm_merge_result (points [total][3], lens [count]), destroy.
 * merge_lines :166-291, merge_2_seqs :67-104, merge_2_reversed_seqs :106-132, helpers :17-65 / :157-164;
 * lm_downsample_seq = downsample_seqs :133-153 (out holds up to n + 1 points, returns the number kept). */
void* lm_merge_create(void);
void lm_merge_destroy(void* merger);
int lm_merge_add_tile(void* merger, const double* points, const int* lens, int n);
long lm_merge_finish(void* merger, long* total_points);
int lm_merge_result(void* merger, double* points, int* lens);
int lm_downsample_seq(const double* seq, int n, double dist_min, double* out);

/* ---- K-Lane "RowRef" head, config 4 (baseline/models/heads/row_shared_not_reduc_ref.py) ------------------------
 * softmax_rows :179-180,239-240 (in place); select :199-204; gather :207-211; scatter :227-230 (shrinking-range quirk);
 * decode :334-363.  Layouts: x [B,H,W,8] NHWC, ext [B,H,L,2], cls [B,H,L,W], tokens on the fixed grid t = b * L + lane,
 * [B*L][8*H*5] in (c h w) order; valid [B][L] = the reference's lane selection (mean existence > thr_ext), computed on the device. */
int lm_softmax_rows(void* stream, float* x, long rows, int cols);
int lm_rowref_select(void* stream, const float* ext, const float* cls, float* mean_out, int* valid, float thr_ext, int* corr,
                     int B, int H, int W, int L);
int lm_rowref_gather(void* stream, const float* x_nhwc8, const int* corr, float* tok, int B, int H, int W, int L);
int lm_rowref_scatter(void* stream, const float* x_nhwc8, const float* tok, const int* corr, const int* valid,
                      float* y_nhwc8, int B, int H, int W, int L);
int lm_rowref_decode(void* stream, const float* ext2, const float* cls2, unsigned char* conf, unsigned char* cls_map,
                     int* col_idx, int B, int H, int W, int L);

/* ---- sparse-voxel LiDAR encoder, config 5 (baseline/models/pcencoder/lidarencoder.py) ----------------------------
 * PARITY UNPINNED for voxelize / sparse convolutions: the reference only instantiates and calls mmdet3d's
 * VoxelizationByGridShape (:29) and SparseEncoder (:33, called :93,:102); their published behaviour is restated.
 * voxelize_hard replaces :104-129 for one sample (hard voxelisation + mean of the kept points + batch index);
 * row_base / row_end are DEVICE ints (row range of this sample in the batch's feats / coords).
 * sparse_grid_build / sparse_conv_outputs / sparse_rulebook / conv_gather_mfma_f32 replace the SparseEncoder call :93
 * (SubMConv3d and SparseConv3d, BatchNorm1d folded, ReLU, residual); ksp_zyx9 = {kz,ky,kx, sz,sy,sx, pz,py,px}.
 * sparse_to_dense_nhwc = SparseConvTensor.dense().view(N, C*D, H, W) + torch.flip(dims=[2]) (:70);
 * upsample_bicubic_nhwc = F.interpolate(mode='bicubic', align_corners=False) (:72). */
long lm_voxelize_workspace_bytes(long n_points);
/* The two device-wide primitives under the voxeliser and the output-site compaction (csrc/prim.hip: the library's own kernels, where
 * mmdet3d's hard_voxelize / spconv's indice generation loop or hash on the device): exclusive prefix sum of u32 (wraps; in == out
 * allowed) and STABLE sort of (u32 key, u32 value) pairs by the low end_bit bits of the keys, in place. */
long lm_scan_workspace_bytes(long n);
int lm_exclusive_scan_u32(void* stream, const unsigned* in, unsigned* out, long n, void* workspace, long workspace_bytes);
long lm_sort_pairs_workspace_bytes(long n);
int lm_sort_pairs_u32(void* stream, unsigned* keys_io, unsigned* vals_io, long n, int end_bit, void* workspace, long workspace_bytes);
int lm_voxelize_hard(void* stream, const float* points, long n, const float* range_lo_xyz, const float* voxel_size_xyz,
                     const int* grid_xyz, int max_points, int max_voxels, int batch_idx, const int* row_base, int cap_rows,
                     float* feats, int ldf, int* coords, int* row_end, int raster_order, void* workspace,
                     long workspace_bytes);
int lm_sparse_grid_build(void* stream, const int* coords, long n, int* grid, int B, int D, int H, int W);
long lm_sparse_conv_outputs_workspace_bytes(long out_cells);
int lm_sparse_conv_outputs(void* stream, const int* in_coords, long n_in, int B, const int* ksp_zyx9, int Do, int Ho, int Wo,
                           int* out_grid, int* out_coords, int cap_rows, int* out_count, void* workspace, long workspace_bytes);
int lm_sparse_rulebook(void* stream, const int* out_coords, long n_out, const int* in_grid, int B, int D, int H, int W,
                       const int* ksp_zyx9, int* nbr);
int lm_conv_gather_mfma_f32(void* stream, const float* x, int ldx, const int* nbr, int taps, const float* wp, int CoutP,
                            const float* scale, const float* shift, const float* res, int ldr, float* y, int ldy,
                            long M, int Cin, int Cout, int act);
int lm_sparse_to_dense_nhwc(void* stream, const float* feats, int ldf, const int* coords, long n, float* out, int B, int D,
                            int H, int W, int C, int flip_h);
int lm_upsample_bicubic_nhwc(void* stream, const float* x, float* y, int B, int H, int W, int C, int Ho, int Wo);

/* ---- LAS ingest (replaces laspy in `read_las`, baseline/datasets/laserlane_proposals.py:618-636; ASPRS LAS 1.0-1.4,
 * point data record formats 0-10, uncompressed).  parse_header: HOST bytes of the file.  decode_points: records = DEVICE
 * copy of the point data (4-byte aligned, padded to a multiple of 4 bytes); out [n][4] f32 = X*scale + (offset - shift)
 * in f64, intensity raw (normalise 0) or (clip(i, lo, hi) - lo) / hi (normalise 1, read_las). */
typedef struct {
    int version_major, version_minor, point_format, record_len;
    long n_points, offset_to_points;
    double scale[3], offset[3], min_xyz[3], max_xyz[3];
} LmLasHeader;
int lm_las_parse_header(const unsigned char* bytes, long len, LmLasHeader* out);
int lm_las_decode_points(void* stream, const unsigned char* records, int record_len, long n, const double* scale,
                         const double* offset, const double* shift, float inten_lo, float inten_hi, int normalise,
                         float* out_xyzi);

/* ---- PNG tile ingest (replaces PIL in `load_img`, baseline/datasets/laserlane_proposals.py:85-98 and laserlane.py:214-219:
 * np.array(Image.open(path)) -> uint8 HWC).  Host code, no third-party library (PIL runs zlib; the DEFLATE decoder here is
 * csrc/inflate.h), 8-bit non-interlaced grey / grey+alpha / RGB / RGBA only; palette, 16-bit and Adam7 files are refused; chunk CRCs
 * and the zlib Adler-32 are verified.  C = channels (1, 2, 3, 4).
 * lm_png_decode_files_u8 inflates n files of identical geometry on `threads` host threads into out [n][H][W][C], the layout
 * lm_tile_ingest_u8 takes.  lm_zlib_inflate is that decoder on a bare zlib stream (RFC 1950), *produced = inflated bytes. */
int lm_png_info(const unsigned char* data, long size, int* H, int* W, int* C);
int lm_png_decode_u8(const unsigned char* data, long size, unsigned char* out_hwc, long out_bytes);
int lm_png_decode_files_u8(const char* const* paths, int n, unsigned char* out_nhwc, int H, int W, int C, int threads);
int lm_zlib_inflate(const unsigned char* data, long size, unsigned char* out, long capacity, long* produced);

/* ---- per-tile polyline JSON (save_lane_seq_2d, baseline/utils/io_utils.py:58-93: json.dump(records, indent=4)) ----------------
 * lane_vertexes [n_lines][row_size][3] doubles = (row, col, semantic), a vertex exists iff col > 0, lines with < 2 vertices are dropped.
 * The text is byte-identical to the reference's (numbers formatted like CPython's float repr).  lm_lane_json_text returns the text
 * length; it writes (NUL-terminated) only if cap is large enough, so call it with out = NULL first.  Host code. */
long lm_lane_json_text(const double* lane_vertexes, int n_lines, int row_size, int with_pervertex_semantics, char* out, long cap);
int lm_lane_json_write(const double* lane_vertexes, int n_lines, int row_size, int with_pervertex_semantics, const char* path);
/* 3-D polylines after the back-projection / merge (save_seqs_json, utils/io_utils.py:11-15, records built at coor_img2pc.py:205-212):
 * seqs [L][Vmax][D] doubles, lens [L] in [1, Vmax]; keys "seq", "seq_len", "init_vertex", "end_vertex"; same text as json.dump(indent=4). */
int lm_seqs_json_write(const double* seqs, const int* lens, int L, int Vmax, int D, const char* path);

#ifdef __cplusplus
}
#endif
#endif /* LANEMAP_HIP_H */
