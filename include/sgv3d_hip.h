/*
 * sgv3d_hip.h — C ABI of libsgv3d_hip.so, the MI355X (gfx950) kernels behind the SGV3D / BEVHeight
 * camera->BEV forward path.
 *
 * Plain C: pointers are DEVICE pointers unless marked "host"; sizes are ints / size_t; `stream` is a
 * hipStream_t passed as void* (NULL = the null stream).  No torch types.  Every entry point only
 * ENQUEUES work on `stream` (no allocation, no synchronisation, graph-capturable) and returns
 * 0 on success or a negative SGV3D_E* code; sgv3d_last_error() gives the message for the calling
 * thread.  Nothing is retained between calls: all buffers, including workspaces, are owned by the
 * caller.
 *
 * Each declaration cites the reference interface it replaces (paths relative to the reference repo).
 */
#ifndef SGV3D_HIP_H
#define SGV3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGV3D_OK 0
#define SGV3D_EINVAL (-1)   /* bad argument (null pointer, non-positive size, unsupported shape) */
#define SGV3D_ELAUNCH (-2)  /* hipLaunchKernel / hipMemsetAsync reported an error                 */
#define SGV3D_ENOSPACE (-3) /* workspace too small                                               */

const char *sgv3d_last_error(void);
/* ABI version, bumped on any signature change. */
int sgv3d_abi_version(void);

/* ================================================================================================
 * Voxel pooling ("splat")
 * ================================================================================================ */

/* Drop-in for voxel_pooling_forward_wrapper + voxel_pooling_forward_kernel_launcher
 *   ops/voxel_pooling/src/voxel_pooling_forward.cpp:26-39, ops/voxel_pooling/src/voxel_pooling_forward_cuda.cu:9-56
 * Same argument order and meaning:
 *   geom_xyz        int32 [B, N, 3]   voxel index (x, y, z) per frustum point
 *   input_features  f32   [B, N, C]
 *   output_features f32   [B, Y, X, C] pre-zeroed by the caller, ACCUMULATED in place (the reference atomicAdds into it)
 *   pos_memo        int32 [B, N, 3]   pre-filled with -1 by the caller; (b, y, x) written for kept
 *                                      points; may be NULL (inference)
 * A launch failure is reported through the return code instead of exit(-1) (..cuda.cu:51-55).
 *
 * This is the symbol the reference's own Python wrapper reaches through voxel_pooling_ext (INTEGRATION.md level 1), so it
 * carries the fast path itself: per (device, stream, sizes) it keeps a voxel plan (below) in memory the library owns.
 * Every call compares geom_xyz with the tensor the plan was built for ON THE DEVICE (the same pass writes pos_memo); while
 * it is unchanged -- a roadside camera -- the rows are summed by the deterministic gather (vp_gather3_kernel, added to
 * output_features, empty voxels untouched); a call whose geom_xyz differs is served by the float-atomic scatter of the
 * reference (order-nondeterministic, like the reference) and the plan is rebuilt by a later call, once the host has seen
 * the device's note -- the host never waits for the device.  Inside a stream capture with no plan yet, with channel counts
 * the gather does not cover (C % 4 != 0, C < 24 or C > 256), with unaligned feature pointers, or with
 * SGV3D_VP_LEVEL1_CACHE=0 the call is the plain scatter.  At most 8 plans are kept (least recently used evicted);
 * sgv3d_voxel_pooling_cache_clear() frees them (synchronises the device), sgv3d_voxel_pooling_cache_stats() reports
 * {calls, calls served through a plan, scatter-only calls, gated plan builds enqueued}. */
int sgv3d_voxel_pooling_forward(int batch_size, int num_points, int num_channels,
                                int num_voxel_x, int num_voxel_y, int num_voxel_z,
                                const int32_t *geom_xyz, const float *input_features,
                                float *output_features, int32_t *pos_memo, void *stream);
/* The same call for a caller that owns the output allocation (this build's Python operator, which replaces
 * ops/voxel_pooling/voxel_pooling.py:37-52): output_features need NOT be zeroed -- every row is written, empty voxels as
 * zeros (when the call falls back to the scatter, the library zeroes the map itself first).  Saves the 4 Y X C byte fill of
 * voxel_pooling.py:37-38.  Same plans, same sums, same pos_memo as sgv3d_voxel_pooling_forward. */
int sgv3d_voxel_pooling_forward_fresh(int batch_size, int num_points, int num_channels,
                                      int num_voxel_x, int num_voxel_y, int num_voxel_z,
                                      const int32_t *geom_xyz, const float *input_features,
                                      float *output_features, int32_t *pos_memo, void *stream);
/* Which gather kernel the planned / level-1 / fused entries launch: 0 or 2 = the voxel-owner kernel (round 4, default),
 * 1 = the slot-balanced kernel of round 3.  Both give exact sums of the same rows; their fixed summation orders differ (the
 * fused lift-splat entry and the operator always use the same one, so they stay bitwise equal).  Also SGV3D_VP_KERNEL=slot
 * at load time.  (Tests and probes; not part of the reference's interface.) */
int sgv3d_voxel_pooling_select_kernel(int which);
/* the kernel those entries launch for these sizes (fused = the lift-splat form): 1 slot-balanced, 2 voxel-owner, 0 = neither
 * (channel counts outside 24 .. 256 / not a multiple of 4 take the generic gather) -- for reports (bench.py) */
int sgv3d_voxel_pooling_kernel_for(int batch_size, int num_points, int num_channels, int num_voxel_x, int num_voxel_y, int fused);
int sgv3d_voxel_pooling_cache_clear(void);
int sgv3d_voxel_pooling_cache_stats(unsigned long long *out4);

/* The float-atomic scatter alone (voxel_pooling_forward_cuda.cu:9-36 as written: one atomicAdd per kept (point, channel),
 * order-nondeterministic), same arguments; owns no memory.  What ops.voxel_pooling.set_mode("atomic") calls. */
int sgv3d_voxel_pooling_forward_atomic(int batch_size, int num_points, int num_channels,
                                       int num_voxel_x, int num_voxel_y, int num_voxel_z,
                                       const int32_t *geom_xyz, const float *input_features,
                                       float *output_features, int32_t *pos_memo, void *stream);

/* Deterministic, atomic-free formulation of the same operator: a CSR "plan" (voxel -> ascending list
 * of point ids) is built from geom_xyz, then every output row is gathered, reduced in registers and
 * written exactly once (empty voxels are written as zeros: no pre-zeroing needed).  The plan only
 * depends on geom_xyz (i.e. on the camera calibration) and can be reused across calls.
 *
 * Plan layout (bytes from sgv3d_voxel_plan_bytes, 16-B aligned by the caller):
 *   int32 seg_start[B*Y*X + 1] | int32 cursor[B*Y*X + 1] | int32 order[B*N] | int32 slot_voxel[B*N + 1]
 *   | int32 scan scratch | cache header (11 x int32) | int32 geom_xyz copy [B*N*3] (cached build only)
 *   | int32 long-run list [1 + B*N/128 + 1]: count, then the voxels holding more than 128 points (slot_voxel carries
 *     ~voxel for their slots: the gather's waves leave them to the long-run workgroups of the same launch)
 * A plan must not be used by two launches at the same time only if one of them rebuilds it; gathers only read it. */
size_t sgv3d_voxel_plan_bytes(int batch_size, int num_points, int num_voxel_x, int num_voxel_y);

/* Build the plan.  pos_memo as above (may be NULL).  sort_segments != 0 makes every voxel's point
 * list ascending (bitwise-reproducible sums); 0 leaves the arrival order of the fill pass. */
int sgv3d_voxel_plan_build(int batch_size, int num_points,
                           int num_voxel_x, int num_voxel_y, int num_voxel_z,
                           const int32_t *geom_xyz, int32_t *pos_memo,
                           void *plan, size_t plan_bytes, int sort_segments, void *stream);

/* Cached build: geom_xyz depends only on the camera calibration, which is static per roadside camera
 * (SURVEY.md §7.3), so the plan is rebuilt only when geom_xyz is not bytewise the tensor it was built
 * for (or the grid / sort mode differs).  The decision is taken ON THE DEVICE: one compare kernel
 * (reads geom_xyz and the plan's copy once, ~12*B*N*2 bytes) sets a flag in the plan's header and the
 * build kernels return immediately when it is clear -- no host synchronisation, stream-ordered,
 * capturable in a hipGraph.  The plan must have been passed to sgv3d_voxel_plan_init once after
 * allocation.  There is no pos_memo here (training uses sgv3d_voxel_plan_build).
 * sgv3d_voxel_plan_stats_offset: byte offset inside the plan of the header
 *   { int32 dirty, diff, ticket, params[7], builds } -- `builds` counts the real rebuilds (tests, profiling). */
int sgv3d_voxel_plan_init(int batch_size, int num_points, int num_voxel_x, int num_voxel_y,
                          void *plan, size_t plan_bytes, void *stream);
int sgv3d_voxel_plan_build_cached(int batch_size, int num_points,
                                  int num_voxel_x, int num_voxel_y, int num_voxel_z,
                                  const int32_t *geom_xyz, void *plan, size_t plan_bytes,
                                  int sort_segments, void *stream);
size_t sgv3d_voxel_plan_stats_offset(int batch_size, int num_points, int num_voxel_x, int num_voxel_y);

/* output_features f32 [B, Y, X, C], fully overwritten.  Replaces the same reference kernel
 * (voxel_pooling_forward_cuda.cu:9-36) when the caller holds a plan.  One launch (vp_gather_fast_kernel: every voxel is
 * summed by one wave, rows written once, empty voxels zero-filled).  `workspace`: unused since round 3 (the round-2 gather
 * staged partial rows there); sgv3d_voxel_pooling_workspace_bytes returns 16 and the argument may be any pointer or NULL. */
size_t sgv3d_voxel_pooling_workspace_bytes(int batch_size, int num_points, int num_channels);
int sgv3d_voxel_pooling_forward_planned(int batch_size, int num_points, int num_channels,
                                        int num_voxel_x, int num_voxel_y,
                                        const void *plan, const float *input_features,
                                        float *output_features, void *workspace, size_t workspace_bytes,
                                        void *stream);
/* bf16 compute mode: input_features are bf16 [B, N, C] (24 <= C <= 256, C % 4 == 0); sums stay f32.  out_bf16_ld == 0:
 * output_features f32 [B, Y, X, C] as above; out_bf16_ld > 0: output_features bf16 [B, Y, X, out_bf16_ld] (a multiple of 4
 * in [C, 2C]), channels C .. out_bf16_ld - 1 written as zeros -- the layout the bf16 BEV trunk reads (its first convolution
 * rounds its input to bf16 anyway and wants a multiple of 32 input channels). */
int sgv3d_voxel_pooling_forward_planned_bf16(int batch_size, int num_points, int num_channels,
                                             int num_voxel_x, int num_voxel_y, const void *plan,
                                             const void *input_features_bf16, void *output_features, int out_bf16_ld,
                                             void *workspace, size_t workspace_bytes, void *stream);

/* Fused lift-splat (optional fast path beyond the operator boundary):
 *   out[b, y, x, :] = sum over plan points p=(d, pixel) of prob[b, d, pixel] * context[b, pixel, :]
 * without materialising the [B, N, C] lifted tensor of layers/backbones/lss_fpn.py:462-466,486.
 *   prob    f32 [B, D, P]  (softmax over D already applied), P = fH*fW, N = D*P, p = d*P + pixel
 *   context f32 [B, P, C]  channel-last;  workspace as for sgv3d_voxel_pooling_forward_planned */
int sgv3d_lift_splat_planned(int batch_size, int num_depth, int num_pixels, int num_channels,
                             int num_voxel_x, int num_voxel_y, const void *plan,
                             const float *prob, const float *context,
                             float *output_features, void *workspace, size_t workspace_bytes,
                             void *stream);

/* The same with the pooled map written as bf16 rows of out_bf16_ld >= C channels (padding channels zeroed), the hand-off
 * sgv3d_voxel_pooling_forward_planned_bf16 offers: sums in f32 (products not rounded to bf16 first), one rounding on the store.
 * context_bf16 != 0: `context` is bf16 [B, P, C] (8-byte aligned rows: C % 4 == 0); products and sums stay f32.  (Half the
 * L2 -> CU bytes, but the launch is bound by its vector instructions and the unpacking costs more than the bytes save: the
 * model passes f32 rows.) */
int sgv3d_lift_splat_planned_bf16out(int batch_size, int num_depth, int num_pixels, int num_channels, int num_voxel_x,
                                     int num_voxel_y, const void *plan, const float *prob, const void *context, int context_bf16,
                                     void *output_bf16, int out_bf16_ld, void *workspace, size_t workspace_bytes, void *stream);

/* VoxelPooling.backward  ops/voxel_pooling/voxel_pooling.py:58-69
 *   grad_output f32 [B, C, Y, X] with element strides (sb, sc, sy, sx)  (the autograd grad is a
 *   permuted view in the reference; strides let both NCHW-contiguous and NHWC-backed grads in)
 *   grad_input  f32 [B, N, C] fully written: grad_output[b, :, y, x] for kept points, 0 otherwise. */
int sgv3d_voxel_pooling_backward(int batch_size, int num_points, int num_channels,
                                 const int32_t *pos_memo, const float *grad_output,
                                 long long sb, long long sc, long long sy, long long sx,
                                 float *grad_input, void *stream);

/* ================================================================================================
 * Geometry (frustum -> voxel indices)
 * ================================================================================================ */

/* Per-camera 4x4 preparation of LSSFPN.get_geometry / height2localtion
 *   layers/backbones/lss_fpn.py:361 (sensor2virtual @ inverse(intrin)), :367 (sensor2ego @
 *   inverse(sensor2virtual)), :390 (inverse(ida)).
 * Inputs f32 [num_cams, 4, 4] each; prep f32 [num_cams, 3, 4, 4] = (ida_inv, combine_virtual,
 * combine_ego).  Fixed-order float32 arithmetic (see oracle/geometry_ref.py::inv4 / mm4). */
int sgv3d_calib_prep(int num_cams, const float *sensor2ego, const float *sensor2virtual,
                     const float *intrin, const float *ida, float *prep, void *stream);

/* Per-point pass of get_geometry + the quantise expression
 *   layers/backbones/lss_fpn.py:350-401 and :487-488.
 *   frustum   f32 [D, fH, fW, 4]            (the registered buffer, lss_fpn.py:293)
 *   prep      f32 [num_cams, 3, 4, 4]       from sgv3d_calib_prep (or the caller's own 4x4 products)
 *   ref_h     f32 [num_cams]                mats_dict['reference_heights']
 *   bda       f32 [num_cams / cams_per_batch, 4, 4] or NULL (lss_fpn.py:394-400)
 *   voxel_coord, voxel_size  host float[3]  (buffers lss_fpn.py:281-288)
 *   geom_xyz  int32 [num_cams, D, fH, fW, 3]   float->int32: truncate, saturate, NaN -> 0
 *   geom_f    f32   [num_cams, D, fH, fW, 3] or NULL (the un-quantised ego-frame points) */
int sgv3d_geometry_voxel_index(int num_cams, int cams_per_batch, int num_depth, int feat_h, int feat_w,
                               const float *frustum, const float *prep, const float *ref_h,
                               const float *bda, const float *voxel_coord /*host*/,
                               const float *voxel_size /*host*/, int32_t *geom_xyz, float *geom_f,
                               void *stream);

/* The reference recomputes the geometry for every batch (lss_fpn.py:478-488) from calibration tensors its data loader creates anew
 * per batch -- for a roadside camera the same numbers every time.  sgv3d_calib_changed decides ON THE DEVICE whether the n (<= 8)
 * calibration tensors (device pointers, byte counts: multiples of 4) differ from `copy` (sum of nbytes bytes, zero-filled before the
 * first call), the values at the last change: changed[0] = 1 and copy := the tensors if they differ or force != 0, else changed[0] = 0.
 * The *_gated entry points take that flag as `run` (device pointer, NULL = always run): with run[0] == 0 the launch returns at once
 * and its outputs keep what the last run wrote -- no host synchronisation anywhere. */
int sgv3d_calib_changed(int n, const void *const *tensors /*host array of device pointers*/, const int32_t *nbytes /*host*/,
                        void *copy, int force, int32_t *changed, void *stream);
int sgv3d_calib_prep_gated(int num_cams, const float *sensor2ego, const float *sensor2virtual, const float *intrin,
                           const float *ida, float *prep, const int32_t *run, void *stream);
int sgv3d_geometry_voxel_index_gated(int num_cams, int cams_per_batch, int num_depth, int feat_h, int feat_w,
                                     const float *frustum, const float *prep, const float *ref_h, const float *bda,
                                     const float *voxel_coord /*host*/, const float *voxel_size /*host*/, int32_t *geom_xyz,
                                     float *geom_f, const int32_t *run, void *stream);

/* ================================================================================================
 * Lift (softmax over height bins  (x)  context)
 * ================================================================================================ */

/* layers/backbones/lss_fpn.py:462-466 + the permute/contiguous of :486,:490.
 *   height_context f32 [B, P, D + C] channel-last HeightNet output (height logits first)
 *   prob    f32 [B, D, P] or NULL   softmax(height logits) (kept for the fused path / training)
 *   lifted  f32 [B, D, P, C] or NULL  = prob[b,d,p] * context[b,p,c]  (== [B,1,D,fH,fW,C] contiguous) */
int sgv3d_lift(int batch_size, int num_pixels, int num_depth, int num_channels,
               const float *height_context, float *prob, float *lifted, void *stream);
/* bf16 compute mode: the same with the lifted tensor written as bf16 (products formed in f32, rounded once; the largest
 * HBM stream of the path shrinks 2x).  num_channels % 4 == 0; read back by sgv3d_voxel_pooling_forward_planned_bf16. */
int sgv3d_lift_bf16(int batch_size, int num_pixels, int num_depth, int num_channels, const float *height_context,
                    float *prob, void *lifted_bf16, void *stream);

/* ================================================================================================
 * Convolution family (MFMA implicit GEMM, fp32 in / fp32 accumulate, NHWC activations)
 * ================================================================================================ */

/* Replaces the cuDNN/cuBLAS convolutions PyTorch dispatches for every nn.Conv2d /
 * nn.ConvTranspose2d of the path (mmdet ResNet, mmdet3d SECONDFPN, HeightNet, CenterHead:
 * layers/backbones/lss_fpn.py:18-250,296-301, layers/heads/bev_height_head.py:75-110).
 *
 * y[n, oh, ow, co_off + co] = act( scale[co] * sum_{kh,kw,ci} x[n, ih, iw, ci_off + ci] * w[co, kh, kw, ci]
 *                                  + bias[co] + residual[n, oh, ow, co] ) * gate[n, co]
 * with ih = oh*stride - pad + kh*dil (zero outside).  Weights are pre-packed by
 * sgv3d_conv_pack_weight into [cout_pad][k_pad] rows, k = (kh*KW + kw)*cin + ci.
 *
 * mode SGV3D_CONV_DECONV: ConvTranspose2d with kernel == stride (SECONDFPN deblocks): the GEMM has
 * cout*ks*ks columns ordered (dy, dx, co) and column (dy,dx,co) of input pixel (ih,iw) is stored at
 * output pixel (ih*ks+dy, iw*ks+dx), channel co. */
typedef struct sgv3d_conv_desc {
    int batch, in_h, in_w, cin;      /* input  [batch, in_h, in_w, x_ld] (x_ld >= cin_off + cin)        */
    int out_h, out_w, cout;          /* output [batch, out_h, out_w, y_ld]                              */
    int kh, kw, stride, pad, dil;
    int x_ld, x_coff;                /* input channel stride per pixel and first channel                */
    int y_ld, y_coff;                /* output channel stride per pixel and first channel (concat)      */
    int res_ld;                      /* residual channel stride per pixel (0 if no residual)            */
    int relu;                        /* 1: max(.,0) after bias/residual                                 */
    int mode;                        /* SGV3D_CONV_NORMAL / _DECONV / _NCHW_OUT                         */
    int deconv_ks;                   /* kernel == stride of the transposed conv (mode DECONV)           */
    int k_pad, cout_pad;             /* packed-weight geometry (from sgv3d_conv_pack_geometry)          */
    int tile;                        /* 0 = heuristic, else SGV3D_TILE_*                                */
    int x_nchw;                      /* reserved, must be 0 (NCHW images go through sgv3d_nchw_to_nhwc) */
    int k_order;                     /* packed-weight k order, must match sgv3d_conv_pack_weight:       */
                                     /* 0: k = tap*cin + ci;  1: k = ((ci/32)*taps + tap)*32 + ci%32    */
    int split_k;                     /* <= 1: one workgroup per tile sums all of K.  s > 1: s workgroups */
                                     /* per tile, partial sums in the workspace, fixed-order reduce      */
} sgv3d_conv_desc;

#define SGV3D_CONV_NORMAL 0
#define SGV3D_CONV_DECONV 1
#define SGV3D_CONV_NCHW_OUT 2   /* y is [batch, y_ld, out_h, out_w] planes (final head maps)        */
#define SGV3D_CONV_GROUP_PLANES 3 /* y is [cout/g][batch*out_h*out_w][g], g = deconv_ks: one NHWC map per
                                    group of g output channels (the per-branch hidden maps of the head) */

#define SGV3D_TILE_128x128 1
#define SGV3D_TILE_128x64 2
#define SGV3D_TILE_64x128 3
#define SGV3D_TILE_64x64 4
/* sgv3d_conv2d_winograd4_forward only: the grouped GEMM on a 32 x 128 tile (rows per position padded to 32 instead of 64) */
#define SGV3D_TILE_32x128 9
/* sgv3d_conv2d_winograd4_forward only: the grouped GEMM on v_mfma_f32_16x16x4_f32 with 48 x 64 workgroup tiles -- rows per
 * position padded to a multiple of 48 (a 54x96 map: 336 tiles exactly, against 384 / 352 for the 64- / 32-row tiles) */
#define SGV3D_TILE_48x64 10
/* F(4x4) position GEMM with f32-accurate products from three bf16 terms per operand (csrc/gemm_x3_grouped.hip): desc.tile =
 * SGV3D_TILE_X3 | variant, variant = {0: 48, 1: 64, 2: 96, 3: 112, 4: 128} rows per workgroup x 128 columns, + 5: x 160 columns;
 * the weights are those of sgv3d_conv_winograd4_pack_weight_x3 and desc.cout_pad their rows per block */
#define SGV3D_TILE_X3 64
/* ... | SGV3D_TILE_MFIRST: the workgroups walk the output-channel tiles of one m-tile back to back (input rows fetched once
 * from HBM) instead of the m-tiles of one channel tile (weight tile shared); same results */
#define SGV3D_TILE_MFIRST 16
/* ... | SGV3D_TILE_OCC5 (with SGV3D_TILE_64x64, f32, channel-chunk-major weights): the five-workgroups-per-CU form of the 64x64
 * tile -- 32 KB of LDS (swizzled rows), one register stage; for small-K layers (csrc/conv_igemm.hip) */
#define SGV3D_TILE_OCC5 32
/* sgv3d_conv2d_winograd_forward only: desc.tile == SGV3D_WINOGRAD_RESIDENT selects the variant that keeps
 * the input patch of all channels in LDS and walks over the cout tiles (cin <= 96, split_k <= 1: the
 * fused CenterHead branch layer); any other value selects the streaming variant. */
#define SGV3D_WINOGRAD_RESIDENT 6
/* ... == SGV3D_WINOGRAD_HALF: workgroups of 64 tiles x 32 channels whose waves split the 16 Winograd positions in two halves
 * (two workgroups per CU; same weights, same results up to the summation order of the output transform) */
#define SGV3D_WINOGRAD_HALF 8

/* Winograd F(4x4, 3x3) in three launches (csrc/conv_wino4.hip): input transform -> grouped f32-MFMA GEMM over the 36
 * positions -> output transform + epilogue.  Executes 1/4 of the direct form's multiplications (F(2x2): 1/2.25); for 3x3 /
 * stride 1 / pad 1 layers with many channels on small maps (it moves 2.25x the activations through the last-level cache).
 * u_packed: 36 blocks of [cout_pad][k_pad] floats (sgv3d_conv_pack_geometry(cin, cout)); block p = 6 i + j holds the 1x1
 * weight (G g G^T)[i][j] ([cout, cin], G the 6x3 F(4x4,3x3) matrix); sgv3d_conv_winograd4_pack_weight makes all 36 in one
 * launch (k = ci in either k order).
 * desc as for sgv3d_conv2d_winograd_forward, NORMAL mode, no gate, no split-K; desc.k_pad / cout_pad describe one block;
 * desc.tile = SGV3D_TILE_64x64 (default) | SGV3D_TILE_64x128 | SGV3D_TILE_32x128 | SGV3D_TILE_48x64 picks the GEMM tile.  workspace: V and M
 * (sgv3d_conv2d_winograd4_workspace_bytes, 16-B aligned).  fp32 error ~1e-5 of the output scale. */
int sgv3d_conv_winograd4_pack_weight(const float *w_src /*[cout, cin, 3, 3]*/, int cout, int cin, int k_pad, int cout_pad,
                                     float *u_packed /*36 x cout_pad x k_pad*/, void *stream);
/* ... as three bf16 planes per element, layout [36][cout_pad][cin_pad / 32][3][32] (cin_pad % 32 == 0, cout_pad % 32 == 0), for
 * desc.tile = SGV3D_TILE_X3 | variant of sgv3d_conv2d_winograd4_forward: every f32 weight of (G g G^T) split exactly into
 * hi + mid + lo bf16 terms.  Same reference layers (layers/backbones/lss_fpn.py:161-250, layers/heads/bev_height_head.py:97-108). */
int sgv3d_conv_winograd4_pack_weight_x3(const float *w_src, int cout, int cin, int cin_pad, int cout_pad, void *u3_packed,
                                        void *stream);

/* Convolution as an implicit GEMM with f32-accurate products from three bf16 terms per operand (csrc/conv_pw_x3.hip): the 1x1 layers
 * of the mmdet ResNet bottlenecks, of HeightNet and of the necks, and the strided 3x3 / 1x1 / patchify layers between the stages
 * (layers/backbones/lss_fpn.py:175-205,296-297; layers/heads/bev_height_head.py:75-78) -- any kernel with at most 32 taps, any
 * stride / padding / dilation, cin % 32 == 0.  Weights: sgv3d_conv_pack_weight_x3 splits w [cout, cin, kh, kw] f32 once into
 * [cout_pad / 16][kh * kw * cin_pad / 32][3][512] bf16 (fragment order, k = ((ci / 32) * taps + tap) * 32 + ci % 32; cin_pad % 32 ==
 * 0, cout_pad % 32 == 0); activations stay f32 tensors and are split on their way into LDS.  desc as for sgv3d_conv2d_forward
 * (NORMAL mode, no gate; folded BN / bias, residual, ReLU, channel window and concat offsets; desc.split_k > 1 with a workspace of
 * sgv3d_conv2d_workspace_bytes(desc): partial sums, fixed-order reduce), desc.cout_pad = rows of
 * the x3 weights, desc.tile = SGV3D_TILE_X3 | variant [| SGV3D_TILE_MFIRST]: variant & 3 = {0: 32, 1: 64, 2: 128} output pixels
 * per workgroup, variant & 4: 64 instead of 128 channels. */
int sgv3d_conv_pack_weight_x3(const float *w_src, int cout, int cin, int kh, int kw, int cin_pad, int cout_pad, void *u3_packed,
                              void *stream);
int sgv3d_conv2d_x3_forward(const sgv3d_conv_desc *desc, const float *x, const void *u3_packed, const float *scale,
                            const float *bias, const float *residual, float *y, void *workspace, size_t workspace_bytes,
                            void *stream);
size_t sgv3d_conv2d_winograd4_workspace_bytes(const sgv3d_conv_desc *desc /*host*/);
int sgv3d_conv2d_winograd4_forward(const sgv3d_conv_desc *desc /*host*/, const float *x, const float *u_packed,
                                   const float *scale, const float *bias, const float *residual, float *y,
                                   void *workspace, size_t workspace_bytes, void *stream);

/* Packed weight geometry for a GEMM with `k` reduction elements and `n` output columns. */
void sgv3d_conv_pack_geometry(int k, int n, int *k_pad, int *n_pad);

/* Repack an OIHW (nn.Conv2d.weight, [cout, cin, kh, kw]) tensor into the kernel's [cout_pad][k_pad]
 * layout.  k_order 0: k = (kh*KW+kw)*cin_pad + ci (any cin_pad % 4 == 0).  k_order 1 (cin_pad % 32 == 0):
 * k = ((ci/32)*KH*KW + kh*KW+kw)*32 + ci%32, i.e. the taps of one 32-channel chunk are adjacent so their
 * overlapping input pixels are fetched back to back (L1/L2 hits).  `transposed` != 0: the source is an
 * nn.ConvTranspose2d.weight [cin, cout, ks, ks] and the packed rows are (dy, dx, co).  cin_pad >= cin
 * pads channels with zeros. */
int sgv3d_conv_pack_weight(const float *w_src, int cout, int cin, int kh, int kw, int cin_pad,
                           int transposed, int k_order, float *w_packed, int k_pad, int cout_pad,
                           void *stream);

/* scale/bias f32 [cout] (NULL = 1 / 0), residual NHWC f32 or NULL, gate f32 [batch, cout] or NULL.
 * workspace: sgv3d_conv2d_workspace_bytes(desc) bytes (0 / NULL when split_k <= 1), scratch.
 * The input tensor and the packed weights must each be smaller than 3.75 GiB (the kernels address them
 * with 32-bit buffer offsets); outputs are not limited. */
size_t sgv3d_conv2d_workspace_bytes(const sgv3d_conv_desc *desc /*host*/);
int sgv3d_conv2d_forward(const sgv3d_conv_desc *desc /*host*/, const float *x, const float *w_packed,
                         const float *scale, const float *bias, const float *residual,
                         const float *gate, float *y, void *workspace, size_t workspace_bytes,
                         void *stream);

/* Same convolution with the multiplications on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 2.5 PFLOP/s dense):
 * x, w_packed, y and the epilogue operands are the same f32 buffers; both operands are rounded to bf16 (nearest even)
 * between the load registers and LDS, sums are accumulated in f32.  Equals sgv3d_conv2d_forward on operands that are
 * bf16-representable, up to the f32 summation order.  The compute dtype of BASELINE configs 3 and 5. */
int sgv3d_conv2d_forward_bf16(const sgv3d_conv_desc *desc /*host*/, const float *x, const float *w_packed,
                              const float *scale, const float *bias, const float *residual,
                              const float *gate, float *y, void *workspace, size_t workspace_bytes,
                              void *stream);

/* bf16 ACTIVATIONS in HBM (bf16 mode, the convolution chains of the image backbone / BEV trunk): the same bf16-MFMA
 * convolution reading and / or writing bf16 tensors.  io_flags bit 0: x is bf16 [.., x_ld] (element offsets as in the
 * desc); bit 1: y AND residual are bf16 -- mode NORMAL or DECONV, no gate, cout / y_ld / y_coff / res_ld multiples of 8; the
 * epilogue (folded BN, residual, ReLU) runs in fp32 and rounds once to bf16.  Scale and bias stay f32; w_packed is the
 * bf16 copy of the packed weights here (same [cout_pad][k_pad] layout, made by sgv3d_conv_weight_to_bf16): the weights are
 * the operand every workgroup of an output column re-reads from L2, 80 % of the load requests of a 1x1 layer. */
int sgv3d_conv_weight_to_bf16(const float *w_packed, int k_pad, int cout_pad, void *w_packed_bf16, void *stream);
int sgv3d_conv2d_forward_bf16io(const sgv3d_conv_desc *desc /*host*/, const void *x, const void *w_packed,
                                const float *scale, const float *bias, const void *residual, const float *gate,
                                void *y, void *workspace, size_t workspace_bytes, void *stream, int io_flags);

/* Same convolution, float32-accurate, on the bf16 matrix cores ("f32x3"): each operand is split exactly into three bf16
 * terms (hi + mid + lo) and a product is the f32 sum of the six partial products of weight >= 2^-16; the neglected
 * terms are below 2^-23 relative, i.e. one f32 rounding per product.  Bit-exact on data that is exact in bf16. */
int sgv3d_conv2d_forward_f32x3(const sgv3d_conv_desc *desc /*host*/, const float *x, const float *w_packed,
                               const float *scale, const float *bias, const float *residual,
                               const float *gate, float *y, void *workspace, size_t workspace_bytes,
                               void *stream);

/* Winograd F(2x2, 3x3) variant of the same operator for 3x3 / stride 1 / dilation 1 / pad 1 layers with
 * cin % 8 == 0 (2.25x fewer multiplies; cuDNN, which the reference's nn.Conv2d dispatches to, uses the
 * same algorithm family for these layers).  Same descriptor, epilogue, modes (NORMAL / NCHW_OUT /
 * GROUP_PLANES), workspace and split_k semantics (split over input-channel steps of 8) as
 * sgv3d_conv2d_forward; desc.k_pad / cout_pad / k_order are ignored, desc.tile selects the variant (below).  Results agree with the direct
 * form to fp32 rounding (different summation order), bit-exact on small-integer data.
 * Weights: sgv3d_conv_winograd_pack_weight transforms an OIHW [cout, cin, 3, 3] tensor (U = G g G^T,
 * computed in double, rounded once) into sgv3d_conv_winograd_weight_floats(cout, cin_pad) floats ordered
 * [cout tile of 64][cin step of 8][16 positions][2][64][4] -- the order the kernel streams them. */
size_t sgv3d_conv_winograd_weight_floats(int cout, int cin_pad);
int sgv3d_conv_winograd_pack_weight(const float *w_src, int cout, int cin, int cin_pad, float *w_packed,
                                    void *stream);
int sgv3d_conv2d_winograd_forward(const sgv3d_conv_desc *desc /*host*/, const float *x, const float *w_wino,
                                  const float *scale, const float *bias, const float *residual,
                                  const float *gate, float *y, void *workspace, size_t workspace_bytes,
                                  void *stream);

/* ================================================================================================
 * Small layers around the convolutions (all NHWC f32)
 * ================================================================================================ */

/* nn.MaxPool2d(kernel 3, stride 2, pad 1) of the mmdet ResNet stem (lss_fpn.py:296). */
int sgv3d_maxpool3x3s2(int batch, int in_h, int in_w, int channels, const float *x, float *y,
                       void *stream);

/* NCHW f32 [B, C, H, W] -> NHWC [B, H, W, c_pad] with zero-padded channels (image ingest). */
/* MaxPool2d(3, 2, 1) on a bf16 NHWC map (channels % 8 == 0). */
int sgv3d_maxpool3x3s2_bf16(int batch, int in_h, int in_w, int channels, const void *x, void *y, void *stream);
int sgv3d_nchw_to_nhwc(int batch, int channels, int h, int w, int c_pad, const float *x, float *y,
                       void *stream);
/* NHWC [B, H, W, ld] channels [coff, coff+C) -> NCHW [B, C, H, W] (outputs handed back to torch). */
int sgv3d_nhwc_to_nchw(int batch, int channels, int h, int w, int ld, int coff, const float *x,
                       float *y, void *stream);

/* nn.AdaptiveAvgPool2d((1,1)) : NHWC [B, H*W, C] -> [B, C]  (ASPP.global_avg_pool, lss_fpn.py:81-86).
 * Deterministic two-stage sum through a caller-owned workspace. */
size_t sgv3d_global_avgpool_workspace_bytes(int batch, int channels);
int sgv3d_global_avgpool(int batch, int pixels, int channels, int x_ld, const float *x, float *y,
                         void *workspace, size_t workspace_bytes, void *stream);

/* y[b, :] = act(scale * (W @ x[b, :]) + bias): small dense layer on [B, K] vectors (Mlp fc1/fc2,
 * SELayer 1x1 convs on [B,C,1,1], ASPP pooled branch; lss_fpn.py:122-159).  W f32 [N, K] row-major.
 * act: 0 none, 1 relu, 2 sigmoid. */
int sgv3d_dense(int batch, int k, int n, const float *x, const float *w, const float *scale,
                const float *bias, int act, float *y, void *stream);
/* ... skipped when run[0] == 0 (device flag of sgv3d_calib_changed; NULL = always run): y keeps its previous contents. */
int sgv3d_dense_gated(int batch, int k, int n, const float *x, const float *w, const float *scale,
                      const float *bias, int act, float *y, const int32_t *run, void *stream);

/* y[b, p, coff + c] = v[b, c] for every pixel p (F.interpolate of a 1x1 map, lss_fpn.py:101-104). */
int sgv3d_broadcast_channels(int batch, int pixels, int channels, int y_ld, int y_coff,
                             const float *v, float *y, void *stream);

/* SELayer gating x * sigmoid(...)  (layers/backbones/lss_fpn.py:155-159): y[b,p,c] = x[b,p,c] * gate[b,c].
 * x, y NHWC f32 [B, P, C] (may alias). */
int sgv3d_scale_channels(int batch, int pixels, int channels, const float *x, const float *gate,
                         float *y, void *stream);

/* Channel-slice copy: y[b,p,0:C] = x[b,p,coff:coff+C], x row stride x_ld, y contiguous [B,P,C]
 * (the context slice of HeightNet's output, lss_fpn.py:464-465). */
int sgv3d_copy_channels(int batch, int pixels, int channels, int x_ld, int x_coff, const float *x,
                        float *y, void *stream);

/* F.interpolate(x, scale_factor=2, mode='bilinear') (align_corners=False), NHWC f32
 * [B,H,W,C] -> [B,2H,2W,C]  (TaskFPN.forward, layers/backbones/bsm_lss_fpn.py:209-212); any C, float4 path when C % 4 == 0. */
int sgv3d_upsample_bilinear2x(int batch, int h, int w, int channels, const float *x, float *y, void *stream);

/* y = a + b * sigmoid(c), n f32 elements: SABlock product plus the TaskFPN residual
 * (layers/backbones/bsm_lss_fpn.py:151-160, 211). */
int sgv3d_add_mul_sigmoid(long long n, const float *a, const float *b, const float *c, float *y, void *stream);

/* Background suppression of BSMLSSFPN._forward_single_sweep (layers/backbones/bsm_lss_fpn.py:524-529),
 * in place on height_context f32 [B, P, ld] that holds depth logits at [0,D) and context at [D,D+ctx):
 * context *= keep, semantic softmax * keep is written at [D+ctx, D+ctx+sem), remaining channels up to ld
 * are zeroed; keep = 0 where softmax(semantic_logits)[0] > background_threshold, else 1. */
int sgv3d_bsm_compose(int batch, int pixels, int num_depth, int context_channels, int semantic_channels,
                      int ld, const float *semantic_logits, int semantic_ld, float background_threshold,
                      float *height_context, void *stream);

/* mmcv DeformConv2dPack (DCNv1) forward as configured at layers/backbones/lss_fpn.py:190-198 -- 3x3, stride 1, pad 1,
 * dilation 1, deform_groups 1, `groups` channel groups, no bias -- in ONE launch: an implicit GEMM whose A operand is the
 * bilinear sample (zero outside the image) of x at p + tap + offset[p][tap], formed on the way into LDS (csrc/dcn_fused.hip);
 * the column tensor of sgv3d_deform_im2col3x3 is never materialised.
 *   x f32 NHWC [B, H, W, C]; offset f32 [B, H, W, off_ld], (dy, dx) of tap t at 2t, 2t+1
 *   w_packed[g]: the group's [out_per_group, 9 * C/groups] matrix (k = tap * C/groups + ci) packed by sgv3d_conv_pack_weight as a
 *                1x1 layer: [cout_pad][k_pad] floats (sgv3d_conv_pack_geometry(9 * C/groups, out_per_group))
 *   y f32 [B, H, W, y_ld]: channels [y_coff, y_coff + groups * out_per_group) written
 * C/groups % 32 == 0, out_per_group % 4 == 0, groups <= 8, 16-byte aligned pointers. */
int sgv3d_deform_conv3x3_forward(int batch, int h, int w, int channels, int groups, int out_per_group, const float *x,
                                 const float *offset, int off_ld, const float *const *w_packed, int k_pad, int cout_pad,
                                 float *y, int y_ld, int y_coff, void *stream);

/* Deformable 3x3 sampling of mmcv DeformConv2dPack (DCNv1, deform_groups=1, stride 1, pad 1, dil 1;
 * lss_fpn.py:190-198): col[b, p, g, tap, cg] = bilinear(x[b, :, :, g*cpg + cg], p + tap + offset).
 *   x      f32 [B, H, W, C] NHWC;  offset f32 [B, H, W, off_ld] with (dy, dx) of tap t at 2t, 2t+1
 *   col    f32 [B, H*W, groups, 9, C/groups] */
int sgv3d_deform_im2col3x3(int batch, int h, int w, int channels, int groups, const float *x,
                           const float *offset, int off_ld, float *col, void *stream);

/* CenterHead second-layer convs fused over all branches (mmdet3d SeparateHead final conv, 3x3,
 * 64 -> c_k, bias; bev_height_head.py:110): hidden f32 [nb][B][H][W][hc] (one NHWC map per branch, as
 * written by sgv3d_conv2d_forward in mode SGV3D_CONV_GROUP_PLANES), weights f32 [sum_c][3][3][hc], bias
 * f32 [sum_c]; branch_of_out int32 [sum_c] maps an output channel to its branch (<= 4 per branch).
 * out f32 NCHW [B, sum_c, H, W]. */
int sgv3d_head_final_conv(int batch, int h, int w, int num_branches, int hidden_ch, int total_out,
                          const float *hidden, const float *weight, const float *bias,
                          const int32_t *branch_of_out, float *out, void *stream);

/* Both layers of all CenterHead branches in one kernel (mmdet3d SeparateHead = [3x3 cin -> 64 + BN + ReLU,
 * 3x3 64 -> c_k + bias] per branch; bev_height_head.py:110): the 64-channel hidden maps stay in LDS and are
 * never written to HBM.  Same results as sgv3d_conv2d_winograd_forward (GROUP_PLANES) followed by
 * sgv3d_head_final_conv up to fp32 summation order (the 9 taps of a border pixel are added block by block in
 * a fixed order: deterministic).
 *   x         f32 NHWC [batch, h, w, x_ld], channels [x_coff, x_coff + cin), cin % 8 == 0, cin <= 64
 *   w1_wino   first-layer weights of all branches, [num_branches*64, cin, 3, 3] packed by
 *             sgv3d_conv_winograd_pack_weight; scale1 / bias1 f32 [num_branches*64] folded BN (NULL = 1 / 0)
 *   w2        f32 [total_out][3][3][64], bias2 f32 [total_out]; out_begin int32 [num_branches + 1] (device):
 *             branch k owns output channels [out_begin[k], out_begin[k+1]), at most 4
 *   out       f32 NCHW [batch, total_out, h, w]
 *   workspace sgv3d_centerhead_branches_workspace_bytes(...) bytes of scratch (ring partial sums) */
size_t sgv3d_centerhead_branches_workspace_bytes(int batch, int h, int w, int total_out);
int sgv3d_centerhead_branches_forward(int batch, int h, int w, int cin, int x_ld, int x_coff, const float *x,
                                      int num_branches, const float *w1_wino, const float *scale1,
                                      const float *bias1, int total_out, const float *w2, const float *bias2,
                                      const int32_t *out_begin, float *out, void *workspace,
                                      size_t workspace_bytes, void *stream);

/* The same contract with the first layers in Winograd F(4x4, 3x3) form (csrc/head_wino4.hip): a quarter of the direct
 * form's multiplications (F(2x2) above: 1 / 2.25); the transformed input of a 16x16 block stays in LDS for all branches.
 * fp32 rounding of F(4x4) is ~1e-5 of the output scale (F(2x2): ~1e-6).  cin must be 64.
 *   u_packed  sgv3d_centerhead_f4_weight_floats(num_branches) floats, filled by sgv3d_centerhead_f4_pack_weight from the
 *             concatenated first-layer weights f32 [num_branches*64, 64, 3, 3] (OIHW)
 *   everything else (and the workspace size) as for sgv3d_centerhead_branches_forward; a branch may own any number of
 *   output channels */
size_t sgv3d_centerhead_f4_weight_floats(int num_branches);
int sgv3d_centerhead_f4_pack_weight(const float *w1, int num_branches, float *u_packed, void *stream);
int sgv3d_centerhead_branches_forward_f4(int batch, int h, int w, int cin, int x_ld, int x_coff, const float *x,
                                         int num_branches, const float *u_packed, const float *scale1,
                                         const float *bias1, int total_out, const float *w2, const float *bias2,
                                         const int32_t *out_begin, float *out, void *workspace,
                                         size_t workspace_bytes, void *stream);

/* The main loop of sgv3d_centerhead_branches_forward_f4 as a plain 3x3 / stride 1 / pad 1 convolution (folded BN / bias,
 * residual, ReLU, NHWC output) for the two shapes that keep the transformed input resident in LDS: cin == 64 with cout a
 * multiple of 64 (mmdet ResNet layer 1, Bottleneck.conv2), or cout == 64 with cin a multiple of 64 (mmdet3d
 * CenterHead.shared_conv, 256 -> 64).  desc: SGV3D_CONV_NORMAL, cin = the (padded) channel count the weights were packed for;
 * tile / split_k are ignored.  u_packed: sgv3d_conv3x3_f4res_weight_floats(cout, cin) floats (0: shape not covered) from
 * sgv3d_conv3x3_f4res_pack_weight(w OIHW [cout, cin_real, 3, 3], ...). */
size_t sgv3d_conv3x3_f4res_weight_floats(int cout, int cin);
int sgv3d_conv3x3_f4res_pack_weight(const float *w, int cout, int cin_real, int cin, float *u_packed, void *stream);
int sgv3d_conv3x3_f4res_forward(const sgv3d_conv_desc *d, const float *x, const float *u_packed, const float *scale,
                                const float *bias, const float *residual, float *y, void *stream);

/* bf16-mode counterpart (BASELINE configs[2] / [4] compute dtype): the same two layers of all branches in one kernel on
 * the bf16 matrix cores, fp32 accumulation, hidden maps kept in LDS as bf16 (csrc/head_bf16.hip).  cin must be 64.
 *   w1_packed  sgv3d_centerhead_bf16_weight_bytes(num_branches) bytes, filled by sgv3d_centerhead_bf16_pack_weight from
 *              the concatenated first-layer weights f32 [num_branches*64, 64, 3, 3] (OIHW)
 *   w2_packed  sgv3d_centerhead_bf16_weight2_bytes(num_branches) bytes, filled by sgv3d_centerhead_bf16_pack_weight2 from
 *              the final-layer weights f32 [total_out, 3, 3, 64] and out_begin
 *   x, scale1, shift1 (folded BN of the first layers; 16-B aligned), bias2, out_begin, out: as for
 *              sgv3d_centerhead_branches_forward
 * No workspace: the one-pixel ring of hidden values a tile needs from its neighbours is recomputed, not exchanged. */
size_t sgv3d_centerhead_bf16_weight_bytes(int num_branches);
int sgv3d_centerhead_bf16_pack_weight(const float *w1, int num_branches, int cin, void *w1_packed, void *stream);
size_t sgv3d_centerhead_bf16_weight2_bytes(int num_branches);
int sgv3d_centerhead_bf16_pack_weight2(const float *w2, const int32_t *out_begin, int num_branches, void *w2_packed,
                                       void *stream);
int sgv3d_centerhead_branches_forward_bf16(int batch, int h, int w, int cin, int x_ld, int x_coff, const float *x,
                                           int num_branches, const void *w1_packed, const float *scale1,
                                           const float *shift1, int total_out, const void *w2_packed, const float *bias2,
                                           const int32_t *out_begin, float *out, void *stream);

/* The same with the shared map given as a bf16 tensor (bf16-activation mode: x_ld / x_coff multiples of 8). */
int sgv3d_centerhead_branches_forward_bf16x(int batch, int h, int w, int cin, int x_ld, int x_coff, const void *x_bf16,
                                            int num_branches, const void *w1_packed, const float *scale1,
                                            const float *shift1, int total_out, const void *w2_packed, const float *bias2,
                                            const int32_t *out_begin, float *out, void *stream);

/* Tools / tests: select_plain(1) runs the single-role variant of the bf16 head kernel (4 waves, phases of a branch one after
 * the other), select_plain(3) the ping-pong variant (two groups of waves half a period apart) instead of the warp-specialised
 * default (0); all three give bitwise the same results; debug_stamps(buf) makes workgroup 0 of the
 * single-role variant write 4 * num_branches + 2 cycle-counter stamps into the device buffer (NULL switches it off). */
void sgv3d_centerhead_bf16_select_plain(int plain);
void sgv3d_centerhead_bf16_debug_stamps(void *buf);

/* bf16-activation twins of the small layers between the convolutions of HeightNet / MSCThead (csrc/act_bf16.hip; bf16
 * compute mode only).  Tensors marked bf16 are NHWC bf16 with channel counts / strides / offsets multiples of 8; arithmetic is
 * float32 as in the f32 functions of the same name, rounded once (nearest even) on the store.
 *   scale_channels_bf16      y[b,p,c] = x[b,p,c] * gate[b,c]                  x, y bf16; gate f32      (lss_fpn.py:155-159)
 *   global_avgpool_bf16      y[b,c] = mean_p x[b,p,c]                         x bf16; y f32; workspace as the f32 function
 *   broadcast_channels_bf16  y[b,p,y_coff+c] = v[b,c]                         v f32; y bf16            (lss_fpn.py:101-104)
 *   upsample_bilinear2x_bf16 F.interpolate(scale_factor=2, bilinear)          x, y bf16                (bsm_lss_fpn.py:210)
 *   add_mul_sigmoid_bf16     y = a + b * sigmoid(c)                           all bf16, n elements     (bsm_lss_fpn.py:151-160, 211)
 *   deform_im2col3x3_bf16    DCNv1 sampling as sgv3d_deform_im2col3x3         x, col bf16; offset f32  (lss_fpn.py:190-198) */
int sgv3d_scale_channels_bf16(int batch, int pixels, int channels, const void *x, const float *gate, void *y, void *stream);
int sgv3d_global_avgpool_bf16(int batch, int pixels, int channels, int x_ld, const void *x, float *y, void *workspace,
                              size_t workspace_bytes, void *stream);
int sgv3d_broadcast_channels_bf16(int batch, int pixels, int channels, int y_ld, int y_coff, const float *v, void *y,
                                  void *stream);
int sgv3d_upsample_bilinear2x_bf16(int batch, int h, int w, int channels, const void *x, void *y, void *stream);
int sgv3d_add_mul_sigmoid_bf16(long long n, const void *a, const void *b, const void *c, void *y, void *stream);
int sgv3d_deform_im2col3x3_bf16(int batch, int h, int w, int channels, int groups, const void *x, const float *offset,
                                int off_ld, void *col, void *stream);

/* bf16-mode 3x3 / stride 1 / pad 1 convolution with the input patch resident in LDS (csrc/conv_patch_bf16.hip): the
 * algorithm the bf16 configs use where the fp32 configs use Winograd -- the BasicBlock / Bottleneck 3x3 layers of
 * HeightNet (layers/backbones/lss_fpn.py:166-198), MSCThead (bsm_lss_fpn.py:185-257), the BEV trunk
 * (layers/heads/bev_height_head.py:75-110) and the image ResNet.  y = relu?(conv(x) * scale + bias + residual).
 *   w_packed  sgv3d_conv3x3_patch_bf16_weight_bytes(cout, cin) bytes, filled by ..._pack_weight from f32 OIHW [cout, cin, 3, 3]
 *   x         NHWC [batch, h, w, x_ld] (channels x_coff .. x_coff + cin), f32 or bf16 (io_flags bit 0)
 *   y         NHWC [batch, h, w, y_ld] (channels y_coff .. y_coff + cout), f32 or bf16 (io_flags bit 1)
 *   residual  NHWC [batch, h, w, res_ld] in the dtype of y, or NULL;  scale / bias  f32 [cout] or NULL (1 / 0)
 * cin must be a multiple of 32, cout of 8; strides and offsets multiples of 8; pointers 16-B aligned.
 * split_k > 1 (small maps): the 32-channel stages of cin are divided over split_k workgroups per tile, raw partial sums go
 * to workspace (4 * split_k * batch * h * w * cout bytes) and are added in fixed order by the split-K reduce kernel of
 * sgv3d_conv2d_forward, which applies the epilogue; split_k <= cin / 32. */
size_t sgv3d_conv3x3_patch_bf16_weight_bytes(int cout, int cin);
/* Tools: debug_stamps(buf) makes workgroup 0 write 5 + 2 * (cin / 32) cycle-counter stamps (start, prologue done, then per
 * 32-channel stage: MFMAs done / stage handed over, the end of the epilogue, and two stamps inside it) into the device
 * buffer; NULL switches it off. */
void sgv3d_conv3x3_patch_bf16_debug_stamps(void *buf);
int sgv3d_conv3x3_patch_bf16_pack_weight(const float *w, int cout, int cin, void *w_packed, void *stream);
int sgv3d_conv3x3_patch_bf16_forward(int batch, int h, int w, int cin, int cout, int x_ld, int x_coff, int y_ld,
                                     int y_coff, int res_ld, int relu, const void *x, const void *w_packed,
                                     const float *scale, const float *bias, const void *residual, void *y,
                                     int io_flags, int split_k, void *workspace, size_t workspace_bytes, void *stream);

/* bf16-mode implicit-GEMM convolution with bf16 activations in and out whose weights stream from L2 in MFMA-fragment order
 * (csrc/conv_dw_bf16.hip, "direct-weight kernel"): the 1x1 layers and strided shortcuts of the ResNet bottlenecks (mmdet ResNet
 * behind layers/backbones/lss_fpn.py:296-301), the 3x3 layers the patch kernel tiles badly, strided / dilated 3x3 layers (ASPP,
 * lss_fpn.py:58-121) and the 7x7 BEV stem (layers/heads/bev_height_head.py:75-110) in the bf16 configs.
 * y = relu?(conv(x) * scale + bias + residual).  Only the activation rows pass through LDS; up to four workgroups per CU overlap
 * each other's load, MFMA and store phases (most 1x1 layers are bound by HBM).  Uses desc: batch, in_h/in_w, out_h/out_w, cin,
 * cout, kh/kw, stride, pad, dil, x_ld/x_coff, y_ld/y_coff, res_ld, relu, mode = SGV3D_CONV_NORMAL, split_k <= 1 and
 *   desc.tile  SGV3D_TILE_DW_<pixels>x<channels> per workgroup: 64x256 | 128x128 | 256x64 (64 pixels per wave: the HBM-bound
 *              layers) | 128x256 | 256x128 (128 pixels per wave: twice the MFMAs per weight fragment, the deep layers)
 *              | 64x256_DEEP | 128x128_DEEP | 64x128 | 64x128_DEEP (small maps: below)
 *   w_packed   sgv3d_conv_dw_bf16_weight_bytes(cout, cin, kh, kw) bytes, filled by ..._pack_weight from f32 OIHW
 *              [cout, cin_w, kh, kw] (cin_w <= cin: the activation's channel count may be padded, the extra columns are zero)
 *   x          NHWC bf16 [batch, in_h, in_w, x_ld];  y  NHWC bf16 [batch, out_h, out_w, y_ld];  residual  bf16 [.., res_ld] or NULL
 * cin must be a multiple of 32, cout of 8; strides and offsets multiples of 8; pointers 16-B aligned.  Results: f32
 * accumulation of bf16 products in ascending (tap, channel) order, one rounding on the store. */
#define SGV3D_TILE_DW_64x256 31
#define SGV3D_TILE_DW_128x128 32
#define SGV3D_TILE_DW_256x64 33
#define SGV3D_TILE_DW_128x256 34
#define SGV3D_TILE_DW_256x128 35
/* ... with activation rows and weight fragments requested two k-chunks ahead instead of one (two workgroups per CU): for the
 * launches that leave about one workgroup per CU, where nothing else covers the memory round trips.  Same results bit for bit. */
#define SGV3D_TILE_DW_64x256_DEEP 36
#define SGV3D_TILE_DW_128x128_DEEP 37
/* 64 pixels x 128 channels per workgroup (a wave owns one 32-channel tile): twice the workgroups of 64x256 on small maps */
#define SGV3D_TILE_DW_64x128 38
#define SGV3D_TILE_DW_64x128_DEEP 39
size_t sgv3d_conv_dw_bf16_weight_bytes(int cout, int cin, int kh, int kw);
int sgv3d_conv_dw_bf16_pack_weight(const float *w, int cout, int cin_w, int cin, int kh, int kw, void *w_packed, void *stream);
int sgv3d_conv_dw_bf16_forward(const sgv3d_conv_desc *desc /*host*/, const void *x, const void *w_packed, const float *scale,
                               const float *bias, const void *residual, void *y, void *stream);
/* The same layer with the k loop split over desc.split_k workgroups per tile (NORMAL mode; split_k <= ceil(kh kw cin / 64)): for maps
 * whose tiles do not fill 256 CUs.  Partial tiles go to `workspace` (sgv3d_conv_dw_bf16_workspace_bytes(desc) bytes, 16-B aligned, f32
 * [split_k][M][cout]), a second kernel adds them in split order and runs the epilogue: deterministic, the k sum associated per split
 * (differs from split_k = 1 by f32 rounding of the partial sums only).  split_k <= 1: identical to sgv3d_conv_dw_bf16_forward. */
size_t sgv3d_conv_dw_bf16_workspace_bytes(const sgv3d_conv_desc *desc /*host*/);
int sgv3d_conv_dw_bf16_forward_splitk(const sgv3d_conv_desc *desc /*host*/, const void *x, const void *w_packed, const float *scale,
                                      const float *bias, const void *residual, void *y, void *workspace, size_t workspace_bytes,
                                      void *stream);

/* Two layers in one launch (conv_dw_bf16_pair_kernel): conv A = desc (k x k, any stride / dilation, cout == 256, folded BN + ReLU per
 * desc.relu) followed by conv B = 1x1 over those 256 channels with cout2 % 256 == 0 outputs, folded BN, residual and ReLU -- conv2 +
 * conv3 of a ResNet layer-3 bottleneck (mmdet Bottleneck behind layers/backbones/lss_fpn.py:296-301; 23 of them in ResNet-101).  The
 * 256-channel map between the layers stays in LDS.  w_packed = ..._pack_weight(w A), w2_packed = ..._pack_weight(w B as [cout2, 256,
 * 1, 1]); x / y / residual bf16 NHWC; y [batch, out_h, out_w, y_ld] at channel y_coff.  Results: bitwise those of
 * sgv3d_conv_dw_bf16_forward called twice (the middle map rounded to bf16 once). */
int sgv3d_conv_dw_bf16_pair_forward(const sgv3d_conv_desc *desc /*host*/, const void *x, const void *w_packed, const float *scale,
                                    const float *bias, int cout2, const void *w2_packed, const float *scale2, const float *bias2,
                                    const void *residual, int res_ld, void *y, int y_ld, int y_coff, int relu2, void *stream);

/* ================================================================================================
 * Box decode + circle NMS  
 * ================================================================================================ */

/* One CenterHead task: BEVHeight.get_bboxes (models/bev_height.py:116-126) -> mmdet3d 0.18.1
 * CenterHead.get_bboxes, CenterPointBBoxCoder.decode and circle_nms, all on the device.
 *   heatmap f32 logits [B, num_class, H, W]; reg [B,2,H,W]; height [B,1,H,W]; dim [B,3,H,W];
 *   rot [B,2,H,W] (sin, cos); vel [B,2,H,W] or NULL.  Every map is addressed as
 *   base + b*batch_stride + c*H*W + pixel, so channel slices of one [B,70,H,W] buffer can be passed.
 *   post_center_range host float[6] or NULL.
 * Outputs (candidates in descending score order, ties by lower flat index):
 *   boxes f32 [B, max_num, 9] = (x, y, z, dim0, dim1, dim2, rot, vx, vy); scores f32 [B, max_num];
 *   labels int32 [B, max_num] (class inside the task); valid u8 [B, max_num] (score/range mask);
 *   keep u8 [B, max_num] (survives circle NMS, at most post_max_size per sample). */
size_t sgv3d_centerpoint_decode_workspace_bytes(int batch, int num_class, int max_num);
int sgv3d_centerpoint_decode(int batch, int num_class, int h, int w, int max_num, const float *heatmap,
                             const float *reg, const float *height, const float *dim, const float *rot,
                             const float *vel, long long batch_stride, float out_size_factor,
                             float voxel_x, float voxel_y, float pc_x, float pc_y, float score_threshold,
                             const float *post_center_range /*host*/, int norm_bbox, float nms_thresh,
                             int post_max_size, void *workspace, size_t workspace_bytes, float *boxes,
                             float *scores, int32_t *labels, unsigned char *valid, unsigned char *keep,
                             void *stream);

/* The same decode for ALL tasks of the head in three launches (grid.y = task) instead of three per task: what
 * BEVHeightHead.get_bboxes calls.  classes_per_task / nms_thresh: host arrays [num_tasks] (<= 16 tasks); heatmap .. vel: host
 * arrays of num_tasks device pointers (vel NULL, or entries NULL, for heads without velocity), every map with the same
 * batch_stride, h, w; outputs stacked [num_tasks, batch, max_num, ...]. */
size_t sgv3d_centerpoint_decode_tasks_workspace_bytes(int batch, int num_tasks, int max_class, int max_num);
int sgv3d_centerpoint_decode_tasks(int batch, int num_tasks, const int32_t *classes_per_task /*host*/, int h, int w, int max_num,
                                   const void *const *heatmap, const void *const *reg, const void *const *height,
                                   const void *const *dim, const void *const *rot, const void *const *vel,
                                   long long batch_stride, float out_size_factor, float voxel_x, float voxel_y, float pc_x,
                                   float pc_y, float score_threshold, const float *post_center_range /*host*/, int norm_bbox,
                                   const float *nms_thresh /*host*/, int post_max_size, void *workspace, size_t workspace_bytes,
                                   float *boxes, float *scores, int32_t *labels, unsigned char *valid, unsigned char *keep,
                                   void *stream);

/* Tail of mmdet3d 0.18.1 CenterHead.get_bboxes (reached from layers/heads/bev_height_head.py:334-405 /
 * models/bev_height.py:116-126): per sample the boxes that survived circle NMS, task after task in candidate order, with
 * z -= h / 2 and the labels offset by the class counts of the earlier tasks -- one launch instead of 18 masked selections
 * (each a host synchronisation) per sample.
 *   boxes f32 [num_tasks, batch, max_num, 9], scores f32 / labels int32 / keep u8 [num_tasks, batch, max_num]: the outputs of
 *   sgv3d_centerpoint_decode for every task, stacked; classes_per_task host int32[num_tasks] (<= 16 tasks)
 *   out_boxes f32 [batch, num_tasks*max_num, 9], out_scores f32 / out_labels int32 [batch, num_tasks*max_num]: the first
 *   counts[b] rows of sample b are valid; counts int32 [batch]. */
int sgv3d_centerpoint_merge_tasks(int batch, int num_tasks, int max_num, const float *boxes, const float *scores,
                                  const int32_t *labels, const unsigned char *keep, const int32_t *classes_per_task /*host*/,
                                  float *out_boxes, float *out_scores, int32_t *out_labels, int32_t *counts, void *stream);

/* ================================================================================================
 * Training-side head functions  
 * ================================================================================================ */

/* Target assignment for every sample and task in one call: BEVHeightHead.get_targets_single
 * (layers/heads/bev_height_head.py:113-253) under mmdet3d 0.18.1 CenterHead.get_targets, gaussian_radius and
 * draw_heatmap_gaussian.
 *   boxes   f32 [batch, n_max, 9] (x, y, z, w, l, h, yaw, vx, vy), padded; labels int32 [batch, n_max], < 0 = padding
 *   classes_per_task  host int32[num_tasks]; label ids run task after task (task 0 owns 0..classes_per_task[0]-1, ...)
 *   h, w    feature map = grid_size // out_size_factor; gaussian_overlap / min_radius / max_objs from train_cfg
 * Outputs (all cleared by the call):
 *   heatmap f32 [batch, total_classes, h, w]  (task t = channels of its classes)
 *   anno_box f32 [num_tasks, batch, max_objs, 10]; ind int64 [num_tasks, batch, max_objs];
 *   mask u8 [num_tasks, batch, max_objs]; slot k of a task = k-th box of the task in the reference's order
 *   (class after class, input order inside a class); boxes past max_objs are dropped as there. */
int sgv3d_centerhead_targets(int batch, int n_max, const float *boxes, const int32_t *labels, int num_tasks,
                             const int32_t *classes_per_task /*host*/, int max_objs, int h, int w, float pc_x,
                             float pc_y, float voxel_x, float voxel_y, float out_size_factor,
                             double gaussian_overlap, int min_radius, int norm_bbox, float *heatmap,
                             float *anno_box, long long *ind, unsigned char *mask, void *stream);

/* Detection loss of ONE task and its gradient with respect to the six prediction maps: BEVHeightHead.loss
 * (layers/heads/bev_height_head.py:255-311) = GaussianFocalLoss(clip_sigmoid(heatmap)) / max(num_pos, 1)
 * + loss_bbox_weight * L1(gathered 10-channel box code, anno_box; mask * code_weights) / max(num, 1e-4).
 *   _stats  stats[0] = count(target_heatmap == 1), stats[1] = sum(mask); the caller averages stats over the
 *           data-parallel ranks (reduce_mean) before passing them on, without a host synchronisation.
 *   maps are addressed as base + b*batch_stride + c*h*w + cell (channel slices of one buffer are fine);
 *   g_* receive d(loss)/d(map) * grad_scale, addressed the same way with grad_batch_stride, or pass all six NULL
 *   for the value only;
 *   loss_out f32[2] = (loss_heatmap, loss_bbox); code_weights host float[10];
 *   workspace sgv3d_centerhead_loss_workspace_bytes(batch) bytes. */
size_t sgv3d_centerhead_loss_workspace_bytes(int batch);
int sgv3d_centerhead_loss_stats(int batch, int num_class, int h, int w, int max_objs, const float *target_heatmap,
                                long long target_batch_stride, const unsigned char *mask, float *stats,
                                void *workspace, size_t workspace_bytes, void *stream);
int sgv3d_centerhead_loss(int batch, int num_class, int h, int w, int max_objs, const float *heatmap,
                          const float *reg, const float *height, const float *dim, const float *rot,
                          const float *vel, long long pred_batch_stride, const float *target_heatmap,
                          long long target_batch_stride, const float *anno_box, const long long *ind,
                          const unsigned char *mask, const float *stats, const float *code_weights /*host*/,
                          float loss_bbox_weight, float grad_scale, float *g_heatmap, float *g_reg,
                          float *g_height, float *g_dim, float *g_rot, float *g_vel, long long grad_batch_stride,
                          float *loss_out,
                          void *workspace, size_t workspace_bytes, void *stream);

/* Weight gradient of sgv3d_conv2d_forward's convolution (mode NORMAL geometry; what nn.Conv2d's backward asks
 * cuDNN for on the training step, exps/...:224-240): dw f32 OIHW [cout, cin, kh, kw] = sum over pixels of
 * dy (NHWC, channel stride y_ld, first channel y_coff) x shifted x (NHWC, x_ld / x_coff).  desc is the FORWARD
 * descriptor; only its geometry and channel strides are read (and desc.tile: 0 = workgroup tile chosen by the
 * library, 1..4 = 64x64, 64x128, 128x64, 128x128 output channels x input channels, for experiments).  split = number of pixel ranges reduced
 * independently (0 = chosen by the library); partial sums are added in a fixed order (deterministic).
 * For an nn.ConvTranspose2d with kernel == stride pass the roles swapped (x := upstream gradient at the fine
 * resolution, dy := the layer's input, desc of the equivalent stride-k convolution): the OIHW result then is
 * the [cin, cout, k, k] layout of ConvTranspose2d.weight. */
size_t sgv3d_conv2d_backward_weight_workspace_bytes(const sgv3d_conv_desc *desc /*host*/, int split);
int sgv3d_conv2d_backward_weight(const sgv3d_conv_desc *desc /*host*/, const float *x, const float *dy, float *dw,
                                 int split, void *workspace, size_t workspace_bytes, void *stream);
/* The same weight gradient with the products on the bf16 matrix cores -- the mixed-precision training mode (the reference
 * trains with --amp_backend native, docs/run_and_eval.md:5,16; BASELINE configs[4] names bf16): x, dy and dw stay f32 tensors,
 * the operands are rounded to bf16 (nearest even) on their way into LDS, accumulation is f32 (v_mfma_f32_32x32x16_bf16), the
 * pixel split and its fixed-order reduce are those of sgv3d_conv2d_backward_weight.  desc.tile: 0 = rule, 1 = 64 x 64,
 * 4 = 128 x 128 (co x ci per workgroup).  Channel counts, strides and offsets % 4 == 0; x / dy 16-byte aligned. */
size_t sgv3d_conv2d_backward_weight_bf16_workspace_bytes(const sgv3d_conv_desc *desc /*host*/, int split);
int sgv3d_conv2d_backward_weight_bf16(const sgv3d_conv_desc *desc /*host*/, const float *x, const float *dy, float *dw, int split,
                                      void *workspace, size_t workspace_bytes, void *stream);
/* ... with ALL NINE TAPS of a 3x3 / stride-1 layer (dilation 1 .. 20) in one workgroup (csrc/conv_wgrad3x3_bf16.hip): a workgroup owns a
 * 64 x 64 (cout x cin) tile and walks down a 32-pixel-wide column of the map, one new dY row segment and ONE new input row per step
 * (the other two stay in an LDS ring), operands kept in NHWC order in LDS and read with ds_read_b64_tr_b16 -- X and dY cross L2 -> LDS
 * once per column instead of nine times.  Same tensors, tolerance and determinism as sgv3d_conv2d_backward_weight_bf16 (different
 * summation order).  split: row chunks per column (0 = rule); workspace ALWAYS needed: ..._alltaps_workspace_bytes(desc, n, split) with
 * n = 1, or the number of problems of the batched form (dy_list / dw_list: HOST arrays of device pointers, one x). */
size_t sgv3d_conv2d_backward_weight_bf16_alltaps_workspace_bytes(const sgv3d_conv_desc *desc /*host*/, int n, int split);
int sgv3d_conv2d_backward_weight_bf16_alltaps(const sgv3d_conv_desc *desc /*host*/, const float *x, const float *dy, float *dw, int split,
                                              void *workspace, size_t workspace_bytes, void *stream);
int sgv3d_conv2d_backward_weight_bf16_alltaps_batched(const sgv3d_conv_desc *desc /*host*/, const float *x, const float *const *dy_list /*host*/,
                                                      float *const *dw_list /*host*/, int n, int split, void *workspace,
                                                      size_t workspace_bytes, void *stream);
/* ... and its batched form (n layers that read the same x, as sgv3d_conv2d_backward_weight_batched). */
size_t sgv3d_conv2d_backward_weight_bf16_batched_workspace_bytes(const sgv3d_conv_desc *desc /*host*/, int n, int split);
int sgv3d_conv2d_backward_weight_bf16_batched(const sgv3d_conv_desc *desc /*host*/, const float *x, const float *const *dy_list /*host*/,
                                              float *const *dw_list /*host*/, int n, int split, void *workspace,
                                              size_t workspace_bytes, void *stream);

/* Batched form of the all-taps kernel (desc.tile 5 layers: 3x3 / stride 1 / dilation 1): n <= 48 weight gradients
 * dw_list[i] = wgrad(x, dy_list[i]) of layers that read the SAME input, in one launch (blockIdx.z = problem).  The 36 first layers of
 * the CenterHead branches (64 -> 64 at 256 x 256, layers/heads/bev_height_head.py:75-110 through mmdet3d SeparateHead) are one 64 x 64
 * tile each: alone a launch needs a pixel split of 256 to fill the chip, batched 36 x 14.  dy_list / dw_list: HOST arrays of device
 * pointers ([batch, out_h, out_w, y_ld] with desc.y_coff / OIHW [cout, cin, 3, 3]).  workspace: ..._batched_workspace_bytes(desc, n, split). */
size_t sgv3d_conv2d_backward_weight_batched_workspace_bytes(const sgv3d_conv_desc *desc /*host*/, int n, int split);
int sgv3d_conv2d_backward_weight_batched(const sgv3d_conv_desc *desc /*host*/, const float *x, const float *const *dy_list /*host*/,
                                         float *const *dw_list /*host*/, int n, int split, void *workspace, size_t workspace_bytes,
                                         void *stream);
/* ... with 1 .. 4 output channels (3x3 / stride 1 / dilation 1, desc.y_ld <= 4: the final layers of the CenterHead branches,
 * layers/heads/bev_height_head.py:75-110 through mmdet3d SeparateHead): a vector-ALU kernel (a lane per input channel, sliding 3x3
 * window in registers, dY broadcast by v_readlane) instead of an MFMA tile that would be 97 % padding; per-wave partial sums in
 * the workspace, added in wave order (deterministic). */
size_t sgv3d_conv2d_backward_weight_thin_workspace_bytes(const sgv3d_conv_desc *desc /*host*/);
int sgv3d_conv2d_backward_weight_thin(const sgv3d_conv_desc *desc /*host*/, const float *x, const float *dy, float *dw, void *workspace,
                                      size_t workspace_bytes, void *stream);
/* The WHOLE backward of n such thin layers in one launch per gradient kind (csrc/conv_thin_grad.hip; the 36 final layers of the
 * CenterHead branches, layers/heads/bev_height_head.py:75-110 through mmdet3d SeparateHead -- the reference asks cuDNN per layer and
 * gradient).  desc: batch, in_h, in_w, cin (multiple of 4), x_ld >= cin (pixel stride of the x / dx tensors: a layer may read / write a channel slice
 * of a wider map, passed as the pointer to its first channel; x_coff == 0), out_h, out_w, pad, kh = kw = 3, stride = dil = 1;
 * cout[i] in 1..4 (host); x_list[i] [batch, in_h, in_w, cin]; dy_list[i] CONTIGUOUS [batch, out_h, out_w, cout[i]]; w_list[i] OIHW
 * [cout[i], cin, 3, 3].  Outputs, each list optional (NULL: that gradient is not computed): dx_list[i] [batch, in_h, in_w, cin]
 * (16-byte aligned), dw_list[i] OIHW, db_list[i] [cout[i]] (= sum of dy over the pixels); single entries of dw_list / db_list may
 * be NULL.  All lists are HOST arrays of device pointers.  Deterministic (fixed-order partial sums).  workspace (weight / bias
 * gradients only): ..._workspace_bytes(desc, n, cout). */
/* ... and their forward (cin <= 64): y_list[i] = conv3x3(x_list[i], w_list[i]) + bias_list[i] in f32 arithmetic whatever the mode,
 * y_list[i] CONTIGUOUS [batch, out_h, out_w, cout[i]]; bias_list or entries of it may be NULL.  The MFMA kernels pad these layers to 64
 * output columns (64 us per 33 MB layer at cfg-2 batch 2); here a layer is one pass over its input at HBM speed. */
int sgv3d_conv3x3_thin_forward_batched(const sgv3d_conv_desc *desc /*host*/, int n, const int32_t *cout /*host*/,
                                       const float *const *x_list /*host*/, const float *const *w_list /*host*/,
                                       const float *const *bias_list /*host*/, float *const *y_list /*host*/, void *stream);
size_t sgv3d_conv3x3_thin_backward_batched_workspace_bytes(const sgv3d_conv_desc *desc /*host*/, int n, const int32_t *cout /*host*/);
int sgv3d_conv3x3_thin_backward_batched(const sgv3d_conv_desc *desc /*host*/, int n, const int32_t *cout /*host*/,
                                        const float *const *x_list /*host*/, const float *const *dy_list /*host*/,
                                        const float *const *w_list /*host*/, float *const *dx_list /*host*/,
                                        float *const *dw_list /*host*/, float *const *db_list /*host*/, void *workspace,
                                        size_t workspace_bytes, void *stream);

/* y[b, oy, ox, :] = x[b, oy/stride, ox/stride, :] where both divide evenly (and stay inside x), else 0.
 * NHWC f32, channels % 4 == 0, out >= (in - 1) * stride + 1.  The data gradient of a strided convolution is a
 * stride-1 convolution of this map with the flipped, transposed weights (sgv3d_amd/conv_grad.py). */
int sgv3d_zero_insert(int batch, int in_h, int in_w, int channels, int stride, int out_h, int out_w,
                      const float *x, float *y, void *stream);

/* out [cin, cout, kh, kw] = w [cout, cin, kh, kw] with the taps rotated by 180 degrees and in / out swapped: the OIHW weights of the
 * stride-1 convolution that computes a data gradient (one launch for torch's flip + transpose + contiguous). */
int sgv3d_weight_rot180_transpose(const float *w, int cout, int cin, int kh, int kw, float *out, void *stream);

/* Every packed weight form of a model refreshed in ONE launch (training: the weights change once per step, the packed forms the
 * kernels read -- implicit-GEMM rows, the rotated / transposed rows of the data gradients, bf16 fragment orders -- are permutations
 * of the parameters with zero padding).  jobs: DEVICE array of n_jobs records of sgv3d_gather_pack_job_bytes() bytes
 *   { const float *src; void *dst; const int32 *idx; int64 n; int32 first_block; int32 bf16; }
 * dst[i] = idx[i] < 0 ? 0 : src[idx[i]] for i < n, rounded to bf16 (nearest even, the pack kernels' rounding) when bf16 != 0;
 * dst and idx 16-byte aligned.  Job k owns the blocks [first_block_k, first_block_{k+1}) with ceil(n / elements_per_block) blocks,
 * first_block ascending from 0; total_blocks = their sum.  The reference repacks nothing (cuDNN reads OIHW); this replaces the
 * per-layer sgv3d_conv_pack_weight / sgv3d_weight_rot180_transpose / ..._bf16_pack_weight launches of a training step. */
int sgv3d_gather_pack_job_bytes(void);
int sgv3d_gather_pack_elements_per_block(void);
int sgv3d_gather_pack(const void *jobs, int n_jobs, int total_blocks, void *stream);

/* One fused AdamW step (torch.optim.AdamW semantics, amsgrad off) over a flat fp32 bucket of n parameters:
 * the optimiser of the reference's configure_optimizers (exps/...:298-305).  grad is multiplied by grad_scale
 * first (1 / world size after a sum all-reduce) and, with clip_coef != NULL, by the DEVICE scalar clip_coef[0]
 * (sgv3d_clip_coef: Lightning's gradient_clip_val, exps/...:405).  step >= 1 is the step count after this update.
 * All four buffers 16-byte aligned. */
int sgv3d_adamw_step(long long n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int step,
                     float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                     const float *clip_coef, void *stream);
/* ... with the step-dependent scalars read from DEVICE memory at kernel start: hyper = f32 [lr, lr / (1 - beta1^step),
 * 1 / sqrt(1 - beta2^step)] -- for a launch recorded in a hipGraph (the recorded step follows the step counter and the learning-rate
 * schedule of exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:298-305 through one sgv3d_adamw_set_hyper launch per
 * replay).  Bitwise sgv3d_adamw_step. */
int sgv3d_adamw_step_dev(long long n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, const float *hyper,
                         float beta1, float beta2, float eps, float weight_decay, float grad_scale, const float *clip_coef,
                         void *stream);
/* hyper[0..2] for sgv3d_adamw_step_dev, computed on the host as sgv3d_adamw_step does and passed as kernel arguments (copied at
 * launch: the caller may stage the next step while this one's update is still queued; no host buffer is shared with the device). */
int sgv3d_adamw_set_hyper(float *hyper, int step, float lr, float beta1, float beta2, void *stream);

/* Global-norm gradient clipping -- Lightning's ``gradient_clip_val=5`` of the reference's Trainer
 * (exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:405; exps/sgv3d/bsm_bev_height_lss_r101_864_1536_256x256.py:529),
 * i.e. torch.nn.utils.clip_grad_norm_(parameters, max_norm, norm_type=2) on the averaged gradients.
 * sgv3d_grad_sumsq: sgv3d_grad_sumsq_partials() doubles per flat bucket (fixed slices, fixed order: bitwise repeatable);
 * sgv3d_clip_coef over the ``count`` partials of ALL buckets: out[0] = min(1, max_norm / (norm + 1e-6)), out[1] = norm, with
 * norm = sqrt(sum) * grad_scale.  Both only enqueue (capturable); the AdamW entries read out[0] on the device. */
int sgv3d_grad_sumsq_partials(void);
int sgv3d_grad_sumsq(long long n, const float *grad, double *partials, void *stream);
int sgv3d_clip_coef(const double *partials, int count, float grad_scale, float max_norm, float *out, void *stream);

/* Training-mode BatchNorm2d over NHWC f32 [pixels, channels] (channels % 4 == 0) fused with the residual add and the
 * ReLU that follow it in the reference's blocks: y = relu(gamma * (x - mean) / sqrt(var + eps) + beta + residual) with
 * the statistics of THIS batch (biased variance), running_mean / running_var updated like nn.BatchNorm2d does
 * (momentum, unbiased variance; NULL = not tracked).  residual NULL = none, relu 0 = none.  save_mean / save_invstd
 * f32 [channels] are kept for the backward.  Backward: dy is the gradient w.r.t. y; dx w.r.t. x, dresidual (NULL = not
 * wanted) w.r.t. the residual, dgamma / dbeta w.r.t. the affine parameters; y (the forward output) supplies the ReLU
 * mask.  Statistics and gradient sums are accumulated in float64 in a fixed order (deterministic).
 * workspace: sgv3d_batchnorm_workspace_bytes(channels) bytes, 16-byte aligned like every tensor here. */
size_t sgv3d_batchnorm_workspace_bytes(int channels);
int sgv3d_batchnorm_train_forward(long long pixels, int channels, const float *x, const float *residual,
                                  const float *gamma, const float *beta, float *running_mean, float *running_var,
                                  float momentum, float eps, int relu, float *y, float *save_mean,
                                  float *save_invstd, void *workspace, size_t workspace_bytes, void *stream);
int sgv3d_batchnorm_train_backward(long long pixels, int channels, const float *x, const float *y, const float *dy,
                                   const float *gamma, const float *save_mean, const float *save_invstd, int relu,
                                   float *dx, float *dresidual, float *dgamma, float *dbeta, void *workspace,
                                   size_t workspace_bytes, void *stream);
/* The backward of y = relu(bn(x)) (no residual) without the forward output: the ReLU mask is bn(x) > 0 recomputed from x with the
 * scale / shift the forward folded from gamma, beta and the saved statistics (the same roundings, so the mask is the forward's):
 * two fewer passes over the map than sgv3d_batchnorm_train_backward.  gamma / beta must be the forward's values. */
int sgv3d_batchnorm_relu_train_backward_from_x(long long pixels, int channels, const float *x, const float *dy, const float *gamma,
                                               const float *beta, const float *save_mean, const float *save_invstd, float *dx,
                                               float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes, void *stream);

/* y[b, 2 i + py, 2 j + px, :] = phases[py * 2 + px][b, i + row0, j + col0, :]: interleaves the four sub-pixel phases of a
 * stride-2 data gradient (each phase is a stride-1 convolution of the upstream gradient with the taps of its parity,
 * sgv3d_amd/conv_grad.py) into the NHWC result.  phases / phase_h / phase_w / row0 / col0: host arrays of 4; channels % 4 == 0. */
int sgv3d_interleave_phases2(int batch, int out_h, int out_w, int channels, const float *const *phases /*host*/,
                             const int32_t *phase_h, const int32_t *phase_w, const int32_t *row0, const int32_t *col0,
                             float *y, void *stream);

/* ---- SGV3D BSM branch, training side ------------------------------------------------------- */

/* get_downsampled_gt_semantic (exps/sgv3d/bsm_bev_height_lss_r101_864_1536_256x256.py:258-276): gt u8 [B, H, W] class
 * ids of the SAM mask image -> labels u8 [B, H/factor, W/factor], the maximum id of each factor x factor block
 * (H, W multiples of factor). */
int sgv3d_semantic_labels_downsample(int batch, int h, int w, int factor, const unsigned char *gt,
                                     unsigned char *labels, void *stream);

/* FocalLoss.forward (losses/focal.py:57-90) over focal_loss_with_logits (losses/_functional.py:37-108; `normalized` and
 * `reduced_threshold` off as in the shipped configs): value and gradient in one pass.
 *   logits f32, element (b, c, p) at b*batch_stride + c*class_stride + p*pixel_stride (floats) -- NHWC rows or NCHW
 *   planes; target_kind 0: u8 labels [B, P], 1: int64 labels [B, P] (multiclass mode: the target of class c is
 *   label == c; labels equal to ignore_index are left out when use_ignore_index), 2: f32 targets addressed like
 *   logits (binary / multilabel mode); alpha < 0 = no alpha weighting; reduction_mean 1: every class is averaged over
 *   the kept pixels and the classes are summed (multiclass) / mean over all elements (target_kind 2), 0: sum.
 *   grad (NULL = not wanted) receives grad_scale * d loss / d logit, addressed like logits; loss_out f32 [1].
 *   workspace: sgv3d_focal_loss_workspace_bytes() bytes. */
size_t sgv3d_focal_loss_workspace_bytes(void);
int sgv3d_focal_loss_with_logits(int batch, int num_classes, int pixels, const float *logits, long long batch_stride,
                                 long long class_stride, long long pixel_stride, const void *target, int target_kind,
                                 float alpha, float gamma, long long ignore_index, int use_ignore_index,
                                 int reduction_mean, float grad_scale, float *grad, float *loss_out, void *workspace,
                                 size_t workspace_bytes, void *stream);

/* Adjoint of sgv3d_upsample_bilinear2x: dy f32 [B, 2H, 2W, C] -> dx f32 [B, H, W, C] (any C). */
int sgv3d_upsample_bilinear2x_backward(int batch, int h, int w, int channels, const float *dy, float *dx, void *stream);

/* Gradients of sgv3d_add_mul_sigmoid (y = a + b * sigmoid(c); da = dy): db = dy * sigmoid(c),
 * dc = dy * b * sigmoid(c) * (1 - sigmoid(c)); n f32 elements, n % 4 == 0. */
int sgv3d_add_mul_sigmoid_backward(long long n, const float *dy, const float *b, const float *c, float *db, float *dc,
                                   void *stream);

/* ---- KITTI-AP evaluator ---------------------------------------------------------------------- */

/* Rotated-box overlaps for every same-image (box, query box) pair of a whole validation set in one launch: replaces
 * rotate_iou_gpu_eval (evaluators/kitti_utils/rotate_iou.py:340-378, the numba.cuda kernel :284-337) and, for
 * box_dim 7, d3_box_overlap (evaluators/kitti_utils/eval.py:153-160).
 *   box_offsets / qbox_offsets i32 [num_images + 1] (device): rows of image m are [off[m], off[m+1]);
 *   tile_offsets i32 [num_images + 1] (device): prefix sum of ceil(N_m/16) * ceil(K_m/16); num_tiles = its last entry;
 *   out_offsets i64 [num_images] (device): out[out_offsets[m] + i*K_m + j] = overlap(box i, query box j) as f32;
 *   boxes / qboxes f64 [*, box_dim] (device): box_dim 5 = (x, y, dx, dy, angle) BEV rectangles; box_dim 7 = camera-frame
 *   boxes (x, y, z, l, h, w, ry), BEV part (x, z, l, w, ry) and height overlap on [y - h, y];
 *   criterion -1: intersection / union, 0: / area of the query box (box_dim 7: volume of the box), 1: the other one,
 *   2: the intersection itself.  The clipping runs in float32 in the reference's operation order. */
int sgv3d_rotate_iou_pairs(int num_images, int num_tiles, const int32_t *box_offsets, const int32_t *qbox_offsets,
                           const int32_t *tile_offsets, const long long *out_offsets, const double *boxes,
                           const double *qboxes, int box_dim, int criterion, float *out, void *stream);

/* HOST function (host pointers, no GPU work): precision / recall / orientation-similarity curves of one (class,
 * difficulty, minimum overlap) cell -- the loop body of eval_class (evaluators/kitti_utils/eval.py:487-556) over
 * compute_statistics_jit (:157-277), get_thresholds (:7-25) and fused_compute_statistics (:289-335).
 *   gt_num / dt_num / dc_num i32 [num_images]; overlaps f64, image after image [dt_num[m]][gt_num[m]];
 *   gt_datas f64 [sum gt][5] (2-D box, alpha); dt_datas f64 [sum dt][6] (2-D box, alpha, score); ignored_gt /
 *   ignored_det i64 (0 evaluate, 1 ignore, -1 other class) and dontcares f64 [sum dc][4] from clean_data (:28-79);
 *   metric 0 bbox / 1 bev / 2 3d; num_valid_gt from clean_data; precision / recall / orientation f64 [41];
 *   num_thresholds (may be NULL) receives the number of recall thresholds found.  Deterministic for any num_threads. */
int sgv3d_kitti_eval_curves(int num_images, const int32_t *gt_num, const int32_t *dt_num, const int32_t *dc_num,
                            const double *overlaps, const double *gt_datas, const double *dt_datas,
                            const int64_t *ignored_gt, const int64_t *ignored_det, const double *dontcares, int metric,
                            double min_overlap, int compute_aos, long long num_valid_gt, int num_threads,
                            double *precision, double *recall, double *orientation, int *num_thresholds);

/* ================================================================================================
 * Training-mode forms of the small layers (csrc/train_misc.hip)
 * ================================================================================================ */

/* nn.MaxPool2d(3, 2, 1) of the image ResNet stem with the arg-max tap (0..8, row-major in the window) kept as one byte per
 * output element, and its adjoint in gather form (deterministic; ties: first maximum in window order, as torch).
 *   x f32 NHWC [B, H, W, C] (C % 4 == 0) -> y f32 [B, OH, OW, C], argmax u8 [B, OH, OW, C]
 *   grad_out f32 [B, OH, OW, C] -> grad_in f32 [B, H, W, C] (fully written) */
int sgv3d_maxpool3x3s2_train_forward(int batch, int in_h, int in_w, int channels, const float *x, float *y,
                                     unsigned char *argmax, void *stream);
int sgv3d_maxpool3x3s2_backward(int batch, int in_h, int in_w, int channels, const unsigned char *argmax,
                                const float *grad_out, float *grad_in, void *stream);

/* Weight gradient of y = x @ w^T (sgv3d_dense): grad_w[n][k] = sum_b grad_out[b][n] * x[b][k]; x [batch, k], grad_out
 * [batch, n], grad_w [n, k].  (The data gradient is sgv3d_dense with the transposed weight.) */
int sgv3d_dense_backward_weight(int batch, int k, int n, const float *x, const float *grad_out, float *grad_w, void *stream);

/* Adjoint of sgv3d_deform_im2col3x3 (mmcv DeformConv2dPack sampling, layers/backbones/lss_fpn.py:190-198):
 *   grad_col f32 in the layout of the forward's `col`  ->  grad_x f32 NHWC [B, H, W, C] (zeroed here, then accumulated
 *   with float atomics like mmcv's deformable_col2im: order-nondeterministic) and grad_offset f32 [B, H, W, grad_off_ld]
 *   (channels 2 t = d/dy, 2 t + 1 = d/dx of tap t; channel sums in fixed order). */
int sgv3d_deform_im2col3x3_backward(int batch, int h, int w, int channels, int groups, const float *x,
                                    const float *offset, int off_ld, const float *grad_col, float *grad_x,
                                    float *grad_offset, int grad_off_ld, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SGV3D_HIP_H */
