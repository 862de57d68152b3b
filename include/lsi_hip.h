/*
 * lsi_hip.h -- C ABI of liblsi_hip.so, the MI355X (gfx950) implementation of
 * the layered-depth-image renderer hot path of google/layered-scene-inference.
 *
 * The reference has no FFI/plugin interface of its own: the boundary it offers
 * is the Python function surface of lsi.geometry.* / lsi.nnutils.helpers
 * (SURVEY.md section 8b).  Each entry point below names the reference
 * function(s) (paths relative to /root/reference) whose arithmetic it replaces;
 * the Python mirror in layered-scene-inference_amd/lsi/ binds them with ctypes
 * (binding shown in INTEGRATION.md).
 *
 * Conventions
 *   - plain C: POD structs, raw device pointers, sizes; no C++/torch types.
 *   - every function returns LSI_OK (0) or a negative LSI_E* code, never throws,
 *     never allocates device memory, never synchronises the host.
 *   - all pointers are DEVICE pointers valid on the device the stream belongs
 *     to; work is enqueued on `stream` (a hipStream_t passed as void*; NULL =
 *     the legacy default stream) and the call returns immediately.
 *   - tensors are fp32.  Inputs carry explicit ELEMENT strides so that both the
 *     reference's channels-last L x B x H x W x C layout and planar (permuted
 *     NCHW) conv outputs are consumed without a copy.  Outputs are contiguous
 *     in the reference's logical shape.
 *   - pixel centres are at +0.5 (helpers.py:107-110); coordinates are (x, y).
 *   - the library keeps no state between calls.
 */
#ifndef LSI_HIP_H_
#define LSI_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSI_VERSION 100 /* 0.1.0 */

/* error codes */
#define LSI_OK 0
#define LSI_EINVAL -1        /* bad shape / stride / flag combination        */
#define LSI_ENULL -2         /* required pointer is NULL                      */
#define LSI_EWORKSPACE -3    /* workspace too small                           */
#define LSI_ELAUNCH -4       /* kernel launch failed (hipGetLastError != 0)   */
#define LSI_EUNSUPPORTED -5  /* valid request this build does not implement   */

/* LsiSplatDesc.flags */
#define LSI_COMPOSE 1u       /* compose_layers=True  (ldi.py:167-171)         */
#define LSI_WANT_DISP 2u     /* compute_trg_disp=True (ldi.py:179-180)        */
#define LSI_HAS_MASK 4u      /* mask pointer is given (else mask == 1)        */
#define LSI_WS_KEEP 8u       /* workspace state is kept between calls (below) */
#define LSI_DETERMINISTIC 16u /* bitwise run-to-run reproducible results: the  */
                             /* STREAM path then adds its partial sums in a   */
                             /* fixed order (slower); TILE is always          */
                             /* reproducible, ATOMIC / ROWBAND never are      */

#define LSI_PACKED_RGBD 32u  /* the caller asserts: tex and disp are views of   */
                             /* ONE buffer of RGBD pixels -- disp == tex + 3   */
                             /* floats, tex_sc == 1, tex_sx == disp_sx == 4,   */
                             /* equal layer / batch / row strides, 16-byte     */
                             /* aligned (what a channels-last conv head with   */
                             /* 4 outputs writes).  One 16-byte load then      */
                             /* fetches a pixel's colour and disparity.  The   */
                             /* entry points verify it (LSI_EINVAL if not).    */

/* LsiSplatDesc.path: which kernel family renders the splat.                  */
#define LSI_PATH_AUTO 0      /* library decides from the descriptor          */
#define LSI_PATH_ATOMIC 1    /* source-parallel, global fp32 atomics; any M   */
#define LSI_PATH_ROWBAND 2   /* target-row-band tiles accumulated in LDS with */
                             /* fp32 LDS atomics; requires M[b][1][3] ==      */
                             /* M[b][2][3] == 0 for every b (target row       */
                             /* independent of disparity) -- lsi_rowband_ok.  */
#define LSI_PATH_STREAM 3    /* row-uniform projections (additionally         */
                             /* M[b][1][0] == M[b][2][0] == 0: rectified      */
                             /* stereo): wave-private LDS row windows updated */
                             /* by plain read-modify-write, merged into an    */
                             /* LDS tile of the band -- lsi_stream_ok.        */

#define LSI_PATH_TILE 4      /* any M: source pixels binned per target tile   */
                             /* with integer LDS exchanges, every target cell */
                             /* summed by its owner thread; no fp32 atomics.  */
                             /* What AUTO resolves to inside lsi_splat_fwd.   */

typedef void* lsi_stream_t; /* hipStream_t */

/*
 * Descriptor of one forward_splat call (ldi.py:71-182).
 *   tex  [l,b,y,x,c]  c in 0..2     element strides tex_s{l,b,y,x,c}
 *   disp [l,b,y,x]                  element strides disp_s{l,b,y,x}
 *   mask [l,b,y,x]   (optional)     element strides mask_s{l,b,y,x}
 * Ht = H * trg_downsampling and Wt = W * trg_downsampling must be integral
 * (the reference only works then: ldi.py:113-125).
 */
/*
 * Caller-owned state of the compact STREAM kernel's adaptive build choice (see
 * lsi_stream_adapt_state below).  Zero-initialise, point ctr_dev at 16 bytes of
 * device memory and ctr_host at 16 bytes of zeroed PINNED host memory, keep both
 * alive as long as the record is used, and do not let two calls that use the
 * same record overlap (one record per device, stream and call geometry).
 */
typedef struct LsiStreamAdapt {
  int32_t state;      /* 0 undecided, 1 twelve waves x 2 sets, 2 sixteen x 1  */
  int32_t pending;    /* a probe's counts are on their way to ctr_host        */
  uint32_t seq;       /* that probe's number                                  */
  uint32_t calls;
  uint32_t* ctr_dev;            /* {folded items, all items, seq}: device     */
  volatile uint32_t* ctr_host;  /* the same three words, pinned host memory   */
} LsiStreamAdapt;

typedef struct LsiSplatDesc {
  int32_t L, B, H, W, Ht, Wt;
  int64_t tex_sl, tex_sb, tex_sy, tex_sx, tex_sc;
  int64_t disp_sl, disp_sb, disp_sy, disp_sx;
  int64_t mask_sl, mask_sb, mask_sy, mask_sx;
  float trg_downsampling; /* s                                              */
  float max_disp;
  float zbuf_scale;       /* |zbuf_scale| <= 160 (else LSI_EINVAL)           */
  float bg_wt;            /* lsi_bg_weight(bg_layer_disp, max_disp, scale)   */
  uint32_t flags;
  int32_t path;
  /* Tuning knobs, 0 = library default: target rows per workgroup, threads   */
  /* per workgroup, and (STREAM) LDS window cells per wave = lsi_stream_ok(). */
  /* reserved: 0.  (tools/ set timing-experiment bits in it; results may then */
  /* be wrong -- see the `dbg` uses in csrc/lsi_splat_stream.hip.)            */
  int32_t tune_rows, tune_threads, tune_window, reserved;
  /* NULL, or the caller's record for the adaptive build choice of the compact */
  /* STREAM kernel (composed output): the library itself keeps no such state.  */
  LsiStreamAdapt* adapt;
} LsiSplatDesc;

/* Library version (LSI_VERSION of the build). */
int lsi_version(void);

/* Human-readable text for an LSI_E* code. */
const char* lsi_strerror(int code);

/*
 * Background weight of the white canvas, ldi.py:115-116:
 *   zbuffer_weights(bg_layer_disp / max_disp, zbuf_scale)   (helpers.py:180-193)
 * with the python-float division done in double, as the reference does.
 * Host function (no GPU work).
 */
float lsi_bg_weight(double bg_layer_disp, double max_disp, double zbuf_scale);

/*
 * Host-side test on a HOST copy of the B projection matrices (row-major 4x4):
 * returns 1 when LSI_PATH_ROWBAND renders this call exactly (target row of a
 * source pixel does not depend on its disparity and the normaliser is positive
 * over the whole source image), else 0.
 */
int lsi_rowband_ok(const LsiSplatDesc* desc, const float* M_host);

/*
 * Host-side test for LSI_PATH_STREAM on a HOST copy of the matrices: returns 0
 * when the path does not apply (projection not row-uniform, unsupported
 * strides/alignment), else the number of LDS window cells per wave the call
 * needs, plus flag bits (put the value in desc->tune_window unchanged).
 * LSI_WANT_DISP is rendered for composed output of unit-normaliser pairs
 * (channels-last or RGBD-pixel textures, no mask, at most 15 layers: the
 * composed view, then a second launch with one tile per layer for the
 * disparity, ldi.py:147-180); LSI_PACKED_RGBD inputs under the same
 * conditions; other requests return 0 (LSI_PATH_TILE renders them).
 */
int lsi_stream_ok(const LsiSplatDesc* desc, const float* M_host);

/*
 * Bytes of device workspace lsi_splat_fwd needs for this descriptor (16-byte
 * aligned; any path).  The ATOMIC path keeps its canvases there, the TILE path a
 * per-view disparity range.  The STREAM
 * path uses it to combine the target rows shared by two neighbouring row bands:
 * a few arrival counters at its start, then partial rows.  The counters must be
 * zero when a call starts and every call leaves them zero; lsi_splat_fwd clears
 * them itself (one small memset on the stream) unless LSI_WS_KEEP promises that
 * the buffer was zero-filled once and has only been used by completed or
 * stream-ordered STREAM-path lsi_splat_fwd calls WITH THE SAME L, B, Ht, Wt AND
 * flags since (the place of the counters depends on those; the other paths
 * write over them).  One workspace must not be used by
 * two calls that can run concurrently.  desc->path AUTO or ATOMIC: the size
 * that serves any path (the ATOMIC canvases dominate); STREAM or TILE: what
 * that path needs (counters, boundary rows, disparity ranges: kilobytes).
 */
size_t lsi_splat_workspace_bytes(const LsiSplatDesc* desc);

/*
 * forward_splat, ldi.py:71-182, with its callees fused into the launch:
 *   projection of every source pixel      helpers.py:116-137 (transform_pts)
 *   divide_safe                           helpers.py:82-85
 *   soft z-buffer weight                  helpers.py:180-193
 *   4-corner bilinear splat x3            sampling.py:171-254
 *   normalise / compose epilogue          ldi.py:157-182
 * M is the B x 4 x 4 src->trg matrix (projection.py:71-86), passed as data.
 * Outputs (contiguous; nl = 1 if LSI_COMPOSE else L):
 *   out_img  [nl,B,Ht,Wt,3]   out_wts [nl,B,Ht,Wt,1] (un-normalised weight sum)
 *   out_disp [nl,B,Ht,Wt,1]   required iff LSI_WANT_DISP
 * mask may be NULL (then LSI_HAS_MASK must be clear).
 */
int lsi_splat_fwd(const LsiSplatDesc* desc, const float* tex, const float* disp,
                  const float* mask, const float* M, float* out_img,
                  float* out_wts, float* out_disp, void* workspace,
                  size_t workspace_bytes, lsi_stream_t stream);

/*
 * Gradient of lsi_splat_fwd w.r.t. tex, disp and mask given the gradients of
 * out_img (g_img, [nl,B,Ht,Wt,3]) and optionally out_wts (g_wts, may be NULL).
 * The reference has no hand-written backward (TF autodiff, train_utils.py:113);
 * this implements the closed form of SURVEY.md section 3.5, reproducing TF's
 * zero-gradient conventions (floor, comparisons, clamps on the closed interval).
 * out_img / out_wts are the forward outputs.  g_tex [L,B,H,W,3], g_disp_in
 * [L,B,H,W], g_mask [L,B,H,W] (NULL unless LSI_HAS_MASK) are contiguous.
 * Gradients through out_disp are not implemented (it feeds no loss:
 * ldi_enc_dec.py:302-357) -- LSI_WANT_DISP is ignored here.
 * workspace: lsi_splat_bwd_workspace_bytes(desc) bytes of device scratch (one
 * float4 per output pixel: the gradient w.r.t. the un-normalised canvases).
 * When desc carries the forward's STREAM verdict for rectified pairs (path
 * LSI_PATH_STREAM + tune_window as lsi_stream_ok returned it), channels-last
 * textures (or RGBD pixels) with W % 4 == 0 (a mask, if any, with unit pixel
 * stride), the streamed kernel
 * (csrc/lsi_splat_bwd_stream.hip) derives that canvas in LDS and leaves the
 * workspace untouched; other descriptors take a pre-pass + one thread per
 * source pixel.
 */
size_t lsi_splat_bwd_workspace_bytes(const LsiSplatDesc* desc);

int lsi_splat_bwd(const LsiSplatDesc* desc, const float* tex, const float* disp,
                  const float* mask, const float* M, const float* out_img,
                  const float* out_wts, const float* g_img, const float* g_wts,
                  float* g_tex, float* g_disp_in, float* g_mask,
                  void* workspace, size_t workspace_bytes, lsi_stream_t stream);

/*
 * forward_splat's two variants of one source view in ONE sweep: the reference's
 * training step renders every LDI twice, once per layer (compose_layers=False)
 * and once composed (compose_layers=True), ldi_enc_dec.py:302-334, repeating
 * identical per-layer splats (ldi.py:129-155).  desc->flags must not contain
 * LSI_COMPOSE or LSI_WANT_DISP.  Outputs: out_img / out_wts [L,B,Ht,Wt,3|1]
 * (per layer) and out_img_c / out_wts_c [1,B,Ht,Wt,3|1] (composed).  The STREAM
 * path sums the layers' tiles in LDS; the other paths render per layer and
 * derive the composed view from those outputs.  Workspace as lsi_splat_fwd.
 */
int lsi_splat_fwd_both(const LsiSplatDesc* desc, const float* tex,
                       const float* disp, const float* mask, const float* M,
                       float* out_img, float* out_wts, float* out_img_c,
                       float* out_wts_c, void* workspace,
                       size_t workspace_bytes, lsi_stream_t stream);

/*
 * Gradient of lsi_splat_fwd_both: the per-layer canvases receive the gradients
 * of both outputs (g_img / g_wts for the per-layer ones, g_img_c / g_wts_c for
 * the composed one; either pair may be NULL).  One gather pass over the source
 * pixels.  Workspace: lsi_splat_bwd_workspace_bytes(desc).
 */
int lsi_splat_bwd_both(const LsiSplatDesc* desc, const float* tex,
                       const float* disp, const float* mask, const float* M,
                       const float* out_img, const float* out_wts,
                       const float* out_img_c, const float* out_wts_c,
                       const float* g_img, const float* g_wts,
                       const float* g_img_c, const float* g_wts_c, float* g_tex,
                       float* g_disp_in, float* g_mask, void* workspace,
                       size_t workspace_bytes, lsi_stream_t stream);

/*
 * Parity/debug view of the projection stage: for every source pixel the four
 * flat target indices (tl,tr,bl,br; x + y*Wt within the batch element,
 * sampling.py:234-241) and the four updates of the weight splat,
 * zbuffer_weight * mask * corner_weight (ldi.py:145-153, sampling.py:213-232).
 *   idx4 [L,B,H*W,4] int32      upd4 [L,B,H*W,4] fp32
 */
int lsi_project_indices(const LsiSplatDesc* desc, const float* disp,
                        const float* mask, const float* M, int32_t* idx4,
                        float* upd4, lsi_stream_t stream);

/*
 * sampling.splat, sampling.py:171-254: 4-corner bilinear forward scatter-add
 * of a C-channel image at float coordinates onto an initial canvas.
 *   src [B,Hs,Ws,C], coords [B,Hs,Ws,2] (x,y), out [B,Ht,Wt,C]; all contiguous.
 * `out` must already hold init_trg_image; updates are added to it in place.
 */
int lsi_splat_generic(int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t Ht,
                      int32_t Wt, const float* src, const float* coords,
                      float* out, lsi_stream_t stream);

/* Gradients of lsi_splat_generic w.r.t. src and coords (g_out [B,Ht,Wt,C]). */
int lsi_splat_generic_bwd(int32_t B, int32_t Hs, int32_t Ws, int32_t C,
                          int32_t Ht, int32_t Wt, const float* src,
                          const float* coords, const float* g_out, float* g_src,
                          float* g_coords, lsi_stream_t stream);

/*
 * projection.forward_projection_matrix (inverse = 0; projection.py:71-86:
 * pad(K_t) [R t; 0 1] pad(K_s^-1)) or inverse_projection_matrix (inverse = 1;
 * :89-106) for B cameras, on the HOST (no device work): k_s, k_t, rot [B,3,3],
 * t [B,3,1], M [B,4,4], all host fp32.  The 3x3 inverse is evaluated in fp64
 * and rounded once; the 4x4 products accumulate sequentially over k with every
 * multiply and add rounded on its own -- the matrices lsi_splat_fwd's
 * bit-exactness contract is stated on (the Python mirror computes the same
 * values with torch ops).
 */
int lsi_projection_matrices(int32_t B, const float* k_s, const float* k_t,
                            const float* rot, const float* t, int32_t inverse,
                            float* M);

/*
 * sampling.batch_scatter_add_tensor, sampling.py:287-313 (and scatter_add_tensor
 * :257-284 with B = 1): out[b, idx[b,i]] += upd[b,i]; duplicates add.
 *   out [B,P] (holds init on entry), idx [B,N] int32 in [0,P), upd [B,N].
 * Out-of-range indices are skipped (TF raises; the Python mirror checks).
 */
int lsi_scatter_add(int32_t B, int64_t P, int64_t N, const int32_t* idx,
                    const float* upd, float* out, lsi_stream_t stream);

/*
 * sampling.bilinear (compose=True), sampling.py:41-132: 4-tap gather with zero
 * outside.  imgs [B,Hs,Ws,C], coords [B,Ht,Wt,2] (x,y), out [B,Ht,Wt,C].
 */
int lsi_bilinear_fwd(int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t Ht,
                     int32_t Wt, const float* imgs, const float* coords,
                     float* out, lsi_stream_t stream);

/*
 * sampling.bilinear (compose=False), sampling.py:124-130: the four taps, each
 * multiplied by its border-validity mask, and the four un-masked bilinear
 * weights, in the reference's order (x0,y0), (x0,y1), (x1,y0), (x1,y1).
 *   taps [4,B,Ht,Wt,C], wts [4,B,Ht,Wt,1].
 * lsi_bilinear_taps_bwd: TF autodiff of that form -- the taps' gradients are
 * scatter-added (through the masks) into g_imgs [B,Hs,Ws,C] (zero on entry; may
 * be NULL), the weights' gradients g_wts [4,B,Ht,Wt,1] (may be NULL: zero) give
 * g_coords [B,Ht,Wt,2] (may be NULL): d wts / d x = (-wy0, -wy1, +wy0, +wy1),
 * d wts / d y = (-wx0, +wx0, -wx1, +wx1); floor / clip / equal carry no gradient,
 * so the taps contribute nothing to g_coords.
 */
int lsi_bilinear_taps(int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t Ht,
                      int32_t Wt, const float* imgs, const float* coords,
                      float* taps, float* wts, lsi_stream_t stream);
int lsi_bilinear_taps_bwd(int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t Ht,
                          int32_t Wt, const float* coords, const float* g_taps,
                          const float* g_wts, float* g_imgs, float* g_coords,
                          lsi_stream_t stream);

/*
 * Gradients of lsi_bilinear_fwd.  g_imgs [B,Hs,Ws,C] must be zero on entry
 * (scatter-add target); g_coords [B,Ht,Wt,2] may be NULL.
 */
int lsi_bilinear_bwd(int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t Ht,
                     int32_t Wt, const float* imgs, const float* coords,
                     const float* g_out, float* g_imgs, float* g_coords,
                     lsi_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Losses on LDIs / rendered views and layer composition: one fused pass     */
/* each (csrc/lsi_loss.hip).  Scalars are DEVICE floats (no host sync);      */
/* gradients come out contiguous in the inputs' logical shape.               */
/* ------------------------------------------------------------------------ */

/* Inputs of zbuffer_composition_loss (lsi/loss/loss.py:66-115), element     */
/* strides: imgs [l,b,y,x,c] c in 0..2, masks/disps [l,b,y,x], trg [b,y,x,c] */
typedef struct LsiLossDesc {
  int32_t L, B, H, W;
  int64_t img_sl, img_sb, img_sy, img_sx, img_sc;
  int64_t mask_sl, mask_sb, mask_sy, mask_sx;
  int64_t disp_sl, disp_sb, disp_sy, disp_sx;
  int64_t trg_sb, trg_sy, trg_sx, trg_sc;
  float bg_layer_disp, max_disp, zbuf_scale;
  int32_t reserved;
} LsiLossDesc;

/* Bytes of device scratch every *_loss_fwd needs (partial sums). */
size_t lsi_loss_workspace_bytes(void);

/*
 * loss.zbuffer_composition_loss, lsi/loss/loss.py:66-115: white background
 * layer at bg_layer_disp appended; p_l = zbuffer_weights(d_l/max_disp)*m_l
 * normalised over the layers; 0.5 * mean(sum_l (img_l - trg)^2 * p_l).
 * masks may be NULL (ones).  out_loss: one device float.
 */
int lsi_zbuf_comp_loss_fwd(const LsiLossDesc* desc, const float* imgs,
                           const float* masks, const float* disps,
                           const float* trg, float* out_loss, void* workspace,
                           size_t workspace_bytes, lsi_stream_t stream);
/* Gradients w.r.t. imgs [L,B,H,W,3], masks [L,B,H,W] (NULL when masks is) and
 * disps [L,B,H,W], scaled by the device scalar g_loss. */
int lsi_zbuf_comp_loss_bwd(const LsiLossDesc* desc, const float* imgs,
                           const float* masks, const float* disps,
                           const float* trg, const float* g_loss, float* g_imgs,
                           float* g_masks, float* g_disps, lsi_stream_t stream);

/*
 * The two regularisers of the disparities in one read of disp [l,b,y,x]:
 *   out2[0] = ldi.disp_smoothness_loss  (lsi/geometry/ldi.py:33-68): mean |d_xx|
 *             + mean |d_xy| + mean |d_yx| + mean |d_yy| of forward differences
 *   out2[1] = loss.decreasing_disp_loss (lsi/loss/loss.py:48-63): mean relu(
 *             d_{l+1} - stop_gradient(d_l)); 0 when L == 1
 */
int lsi_disp_reg_loss_fwd(int32_t L, int32_t B, int32_t H, int32_t W,
                          int64_t sl, int64_t sb, int64_t sy, int64_t sx,
                          const float* disp, float* out2, void* workspace,
                          size_t workspace_bytes, lsi_stream_t stream);
/* g_disp [L,B,H,W] = g2[0] * d out2[0]/d disp + g2[1] * d out2[1]/d disp. */
int lsi_disp_reg_loss_bwd(int32_t L, int32_t B, int32_t H, int32_t W,
                          int64_t sl, int64_t sb, int64_t sy, int64_t sx,
                          const float* disp, const float* g2, float* g_disp,
                          lsi_stream_t stream);

/*
 * View-synthesis loss of the training script, ldi_enc_dec.py:337-357: AREA
 * resize of target [B,H,W,3] (element strides) to Ht x Wt (integer factors),
 * mean_c |t - r_l|, min over the nl layers of recons [nl,B,Ht,Wt,3]
 * (contiguous), crop x_min / y_min pixels on every side, mean.
 */
int lsi_view_synth_loss_fwd(int32_t nl, int32_t B, int32_t Ht, int32_t Wt,
                            int32_t H, int32_t W, int32_t x_min, int32_t y_min,
                            const float* recons, const float* target,
                            int64_t t_sb, int64_t t_sy, int64_t t_sx,
                            int64_t t_sc, float* out_loss, void* workspace,
                            size_t workspace_bytes, lsi_stream_t stream);
/* g_recons [nl,B,Ht,Wt,3]; reduce_min's gradient is split among tied layers. */
int lsi_view_synth_loss_bwd(int32_t nl, int32_t B, int32_t Ht, int32_t Wt,
                            int32_t H, int32_t W, int32_t x_min, int32_t y_min,
                            const float* recons, const float* target,
                            int64_t t_sb, int64_t t_sy, int64_t t_sx,
                            int64_t t_sc, const float* g_loss, float* g_recons,
                            lsi_stream_t stream);

/*
 * layers.compose, lsi/geometry/layers.py:29-70 with helpers.soft_z_buffering
 * (lsi/nnutils/helpers.py:140-160): white background layer at min_disp,
 * per-pixel softmax of log(mask + 1e-8) - depth / temp over the L + 1 layers,
 * hard (first arg-max of the probabilities) or soft blend.
 * imgs [L,N,C], masks [L,N], dmaps [L,N], out [N,C]; all contiguous.
 */
int lsi_compose_fwd(int32_t L, int64_t N, int32_t C, const float* imgs,
                    const float* masks, const float* dmaps, int32_t soft,
                    float min_disp, float depth_softmax_temp, float* out,
                    lsi_stream_t stream);
/* layers.compose_depth, layers.py:73-115.  dmax = max(relu(dmaps), min_disp)
 * over the whole tensor (used when bg_layer != 0).  out [N]. */
int lsi_compose_depth_fwd(int32_t L, int64_t N, const float* masks,
                          const float* dmaps, int32_t bg_layer, float dmax,
                          float min_disp, float depth_softmax_temp, float* out,
                          lsi_stream_t stream);

/*
 * Fused batch norm (batch statistics) + beta + ReLU of the reference's conv
 * layers: slim.batch_norm(center=True, scale=False, epsilon, is_training=True)
 * followed by tf.nn.relu (nets.py:44-67, 95-111, 265-347).  x, y, dy, dx:
 * npix x C values, C innermost (torch channels_last), fp32 (bf16 = 0) or
 * bfloat16 (bf16 = 1); C a multiple of 4 (fp32) / 8 (bf16) with C / that a
 * power of two <= 256, C <= 2048 -- else LSI_EINVAL.  Statistics in fp32
 * (shifted sums, fp32 device atomics across workgroups), biased variance
 * (tf.nn.moments).  relu = 0: batch norm only.
 *   groups: the npix x C arrays are `groups` such blocks stored one after the
 *   other (sub-batches along N), each normalised with its own statistics -- the
 *   reference runs its network once per view, so a batch holding both views
 *   keeps two sets (one launch per pass for all groups).
 *   workspace: lsi_bn_workspace_floats(npix, C, bf16, groups) floats,
 *   ZERO-FILLED ONCE by the caller and then reusable by any later call on the
 *   same stream (the library leaves its counters and accumulators zero).
 *   mean_rstd: [groups][2][C] out (forward), in (backward).  dbeta: [C] out
 *   (summed over the groups).
 */
size_t lsi_bn_workspace_floats(int64_t npix, int32_t C, int32_t bf16, int32_t groups);
int lsi_bn_relu_fwd(const void* x, void* y, const float* beta, float* workspace,
                    float* mean_rstd, int64_t npix, int32_t C, int32_t bf16,
                    int32_t relu, float eps, int32_t groups, lsi_stream_t stream);
int lsi_bn_relu_bwd(const void* x, const void* dy, const float* mean_rstd,
                    const float* beta, void* dx, float* dbeta, float* workspace,
                    int64_t npix, int32_t C, int32_t bf16, int32_t relu,
                    int32_t groups, lsi_stream_t stream);
/* lsi_bn_relu_fwd behind a producer that left the sums of x and x * x in
 * `workspace` (lsi_conv2d_fwd_bnstats / lsi_conv2d_bwd_data_bnstats on the same
 * stream): one pass -- y = relu((x - mean) * rstd + beta), mean_rstd out as from
 * lsi_bn_relu_fwd --, accumulators cleared for the next producer. */
int lsi_bn_relu_norm(const void* x, void* y, const float* beta, float* workspace,
                     float* mean_rstd, int64_t npix, int32_t C, int32_t bf16, int32_t relu,
                     float eps, int32_t groups, lsi_stream_t stream);
/* The hand-over between a producer of statistics and lsi_bn_relu_norm is checked
 * on the device: the producer leaves a tag (C, groups) next to its sums, which
 * lsi_bn_relu_norm requires and clears, and which lsi_bn_relu_fwd / _bwd require
 * to be absent.  A call that finds the workspace in the other state writes NaN
 * (y, mean_rstd / dx, dbeta) instead of numbers computed from somebody else's
 * sums, and leaves the workspace clean.  lsi_bn_stats_discard drops statistics
 * whose consumer will not run (an error between the two calls): the first
 * 16 + 4096 floats of every group's block are zeroed on the stream. */
int lsi_bn_stats_discard(float* workspace, int32_t groups, lsi_stream_t stream);

/*
 * 3x3 stride-1 SAME convolution over 32 input channels on the matrix cores
 * (v_mfma_f32_16x16x32_bf16, fp32 accumulation): the full-resolution layers of
 * the LDI heads.
 *   mode 0  `upcnv1b` (nets.py:104-111): cout = 16 or 32, output bf16
 *           N x H x W x cout (before batch norm);
 *   mode 1  `pred_l` (nets.py:150-158): cout <= 4, bias, sigmoid, output fp32
 *           N x H x W x 4 -- RGBD pixels; scale3 must be 1 (LSI_EINVAL
 *           otherwise: lsi_conv3x3_pred_bwd reads sigmoid' off the stored
 *           output; the caller multiplies the disparities by max_disp);
 *   mode 2  the data gradient of mode 0 with cout = 32: the same convolution
 *           with the kernel transposed and flipped (x = the incoming gradient).
 *   x: bf16 N x H x W x 32, channels innermost (torch channels_last), 16-byte
 *   aligned; W % 16 == 0.  weight: the layer's fp32 parameter cout x 32 x 3 x 3
 *   (rounded to bf16 in the kernel, as torch.autocast does).  bias: [cout]
 *   fp32 or NULL (mode 1).
 */
int lsi_conv3x3_c32_fwd(int32_t N, int32_t H, int32_t W, int32_t cout, int32_t mode,
                        const void* x, const float* weight, const float* bias,
                        float scale3, void* out, lsi_stream_t stream);

/*
 * Backward of the prediction head `pred_l` (nets.py:150-158; TF autodiff of
 * conv + bias + sigmoid): y = sigmoid(conv(x, W) + b), cout <= 4.
 *   g, y: fp32 N x H x W x 4 (the incoming gradient and the forward's output,
 *   RGBD pixels; channels >= cout ignored); x: bf16 N x H x W x 32 (needed for
 *   g_wb); weight: fp32 cout x 32 x 3 x 3.
 *   g_x (may be NULL): bf16 N x H x W x 32, the data gradient (MFMA, K = 9 taps
 *   x 4 channels).  g_wb (may be NULL): fp32 [cout * 288 + cout], weight
 *   gradient cout x 32 x 3 x 3 followed by the bias gradient (matrix cores, K =
 *   pixels, sigmoid'(z) * g as bf16 hi + lo); written (not added to).
 *   sigmoid'(z) * g is formed in fp32 registers and never written.
 *   workspace (needed with g_wb): lsi_conv3x3_pred_bwd_workspace_bytes(N, H, W)
 *   bytes, 16-byte aligned -- the pixel blocks' partial sums, folded by a second
 *   kernel (LSI_EWORKSPACE if smaller).
 */
size_t lsi_conv3x3_pred_bwd_workspace_bytes(int32_t N, int32_t H, int32_t W);
int lsi_conv3x3_pred_bwd(int32_t N, int32_t H, int32_t W, int32_t cout, const float* g,
                         const float* y, const void* x, const float* weight, void* g_x,
                         float* g_wb, void* workspace, size_t workspace_bytes,
                         lsi_stream_t stream);

/*
 * Weight gradient of a 3x3 stride-1 SAME convolution on channels-last bf16
 * tensors (reference nets.py:104-111, the `upcnv*b` layers of the LDI heads; TF
 * autodiff of slim.conv2d):
 *   g_weight[co][ci][ky][kx] = sum over n, y, x of
 *       gy[n][y][x][co] * x[n][y + ky - 1][x + kx - 1][ci]        (zero outside)
 *   x: bf16 N x H x W x cin, gy: bf16 N x H x W x cout (16-byte aligned),
 *   g_weight: fp32 cout x cin x 3 x 3, written (not added to).  cin and cout
 *   multiples of 32 (LSI_EUNSUPPORTED otherwise).  MFMA with K = pixels,
 *   operands transposed by the LDS transpose read; fp32 accumulation.
 *   workspace: lsi_conv3x3_wgrad_workspace_bytes(...) bytes, 16-byte aligned
 *   (partial sums of the pixel blocks; LSI_EWORKSPACE if smaller).
 */
size_t lsi_conv3x3_wgrad_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t cin,
                                         int32_t cout);
int lsi_conv3x3_wgrad(int32_t N, int32_t H, int32_t W, int32_t cin, int32_t cout,
                      const void* x, const void* gy, float* g_weight, void* workspace,
                      size_t workspace_bytes, lsi_stream_t stream);

/*
 * The compact STREAM kernel has two builds (12 waves x two items in flight, 16 x
 * one); which is faster depends on the disparity field, which the planner does
 * not see.  With LsiSplatDesc.adapt set, the kernel counts the items that took
 * its folded routes on a few probe launches (the first calls with the record,
 * then two of every 64); a later call reads the count -- asynchronously copied
 * to the record's pinned host memory, never waited for -- and picks the build.
 * All of that state is the caller's record; with adapt == NULL a call is a pure
 * function of its descriptor (the planner's choice: 12 x 2 for large launches).
 * tune_threads != 0, LSI_S2_WIDE and LSI_S2_ADAPT=0 switch the mechanism off;
 * launches under stream capture use the standing decision and probe nothing.
 * Both builds render the same pixels (both are parity-tested); only timing
 * depends on the record.
 * lsi_stream_adapt_state: the record's standing decision after looking for a
 * pending probe's counts: 0 undecided, 1 twelve waves, 2 sixteen waves, -1 NULL.
 */
int lsi_stream_adapt_state(LsiStreamAdapt* a);

/*
 * Convolutions of the encoder-decoder and the LDI heads on the matrix cores
 * (reference nets.py:29-70 encoder, 73-114 decoder_simple, 244-348 U-Net:
 * slim.conv2d k x k stride 1 | 2 with TF `SAME` padding, slim.conv2d_transpose
 * 4 x 4 stride 2; TF autodiff for the data gradients) as implicit GEMMs
 * (v_mfma_f32_16x16x32_bf16, fp32 accumulation) on bf16 channels-last tensors.
 * LsiConvDesc describes the FORWARD convolution
 *   out[n][oy][ox][co] = sum x[n][oy*stride + ky - pad_t][ox*stride + kx - pad_l][ci]
 *                            * weight[co][ci][ky][kx]            (zero outside)
 *   x: bf16 N x H x W x Cin, out: bf16 N x OH x OW x Cout (16-byte aligned),
 *   weight: the layer's fp32 parameter Cout x Cin x KH x KW (rounded to bf16 on
 *   the way into the kernel's operand order, as torch.autocast rounds it: see
 *   lsi_conv2d_pack below).
 *   Cin, Cout multiples of 32; KH, KW <= 7; stride 1 or 2; pad < kernel size
 *   (LSI_EUNSUPPORTED otherwise: lsi_conv2d_supported() tells beforehand).
 * lsi_conv2d_bwd_data: gx[n][iy][ix][ci] = the transpose of the above applied
 *   to gy (N x OH x OW x Cout) -- for stride 2 as four stride-1 sums over the
 *   sub-kernels of the input pixels' parity classes, no zero-stuffed tensor.
 * A transposed convolution (torch ConvTranspose2d(Cin_T, Cout_T, k, stride,
 * padding p), weight Cin_T x Cout_T x k x k, input N x h x w x Cin_T) IS the
 * data gradient of the forward convolution {H = stride h, W = stride w,
 * Cin = Cout_T, OH = h, OW = w, Cout = Cin_T, pad_t = pad_l = p} with the same
 * weight memory: its forward is lsi_conv2d_bwd_data of that descriptor, its
 * data gradient lsi_conv2d_fwd.
 * The weights enter in the kernels' operand order, bf16 [tap][out ch][in ch]:
 * lsi_conv2d_pack(d, mode, weight, packed, ...) writes it from the layer's fp32
 * parameter -- mode 0 for lsi_conv2d_fwd, mode 1 for lsi_conv2d_bwd_data (roles
 * of the channels swapped, taps in parity-class order) -- into
 * lsi_conv2d_packed_bytes(d) bytes (16-byte aligned; LSI_EWORKSPACE if
 * smaller).  mode | 2: the parameter is stored with torch's channels-last
 * strides (Cout x KH x KW x Cin in memory: what module.to(memory_format=
 * torch.channels_last) leaves) instead of contiguously.  A caller whose weights
 * have not changed may keep the packed form -- and one whose optimiser updates
 * them has to pack again after every update (torch's fused optimisers do not
 * move a parameter's version counter: a stale pack is silent).
 */
typedef struct LsiConvDesc {
  int32_t N, H, W, Cin;    /* input  N x H x W x Cin   */
  int32_t OH, OW, Cout;    /* output N x OH x OW x Cout */
  int32_t KH, KW, stride;  /* kernel size; 1 or 2       */
  int32_t pad_t, pad_l;    /* zero rows / columns before the input (TF SAME: total / 2) */
} LsiConvDesc;
int lsi_conv2d_supported(const LsiConvDesc* d);
size_t lsi_conv2d_packed_bytes(const LsiConvDesc* d);
int lsi_conv2d_pack(const LsiConvDesc* d, int32_t mode, const float* weight, void* packed,
                    size_t packed_bytes, lsi_stream_t stream);
/* Many layers packed by ONE launch (a training step re-packs every layer after
 * the optimiser's update: one launch instead of two per layer):
 * lsi_conv2d_pack_job fills a host-side job record for (d, mode, weight, packed)
 * and the number of workgroups it needs; the caller sets block0 of every record
 * to the running sum, copies the table to the device and passes it to
 * lsi_conv2d_pack_many with the total. */
typedef struct LsiPackJob {
  const float* w;     /* the layer's parameter Cout x Cin x KH x KW          */
  void* dst;          /* packed bf16 weights                                  */
  int32_t D0, D1, khw, tr, ntaps, block0;
  int8_t tap[56];     /* ky * KW + kx of every tap, in the kernel's order     */
} LsiPackJob;
int lsi_conv2d_pack_job(const LsiConvDesc* d, int32_t mode, const float* weight, void* packed,
                        size_t packed_bytes, LsiPackJob* job, int32_t* nblocks);
int lsi_conv2d_pack_many(const LsiPackJob* jobs_device, int32_t njobs, int32_t total_blocks,
                         lsi_stream_t stream);
int lsi_conv2d_fwd(const LsiConvDesc* d, const void* x, const void* packed, void* out,
                   lsi_stream_t stream);
int lsi_conv2d_bwd_data(const LsiConvDesc* d, const void* gy, const void* packed, void* gx,
                        lsi_stream_t stream);
/*
 * The same two calls for a layer that slim.batch_norm follows (every
 * slim.conv2d / conv2d_transpose of nets.py:44-67, 95-111, 265-347): the
 * kernel's epilogue also adds the sums of y and y * y of its (bf16-rounded)
 * output -- per channel and sub-batch group, what lsi_bn_relu_fwd's first pass
 * would read the tensor back for -- to the accumulators of `bn_workspace`
 * (lsi_bn_workspace_floats; zero-filled once).  The caller completes the layer
 * with lsi_bn_relu_norm(out, y, ...) as the NEXT call that uses this workspace
 * on the stream: it folds the sums, forms mean / rstd and clears the
 * accumulators.  groups must divide N.  (Plain sums, no shift: a layer whose
 * |mean| is 10^3 times its standard deviation loses the variance -- not a
 * convolution without bias behind a batch norm.)
 */
int lsi_conv2d_fwd_bnstats(const LsiConvDesc* d, const void* x, const void* packed, void* out,
                           float* bn_workspace, int32_t groups, lsi_stream_t stream);
int lsi_conv2d_bwd_data_bnstats(const LsiConvDesc* d, const void* gy, const void* packed,
                                void* gx, float* bn_workspace, int32_t groups,
                                lsi_stream_t stream);
/*
 * A convolution behind a skip connection (tf.concat([a, b], axis=3) in front of
 * slim.conv2d: nets.py:104-106 `upcnv{n}b`, :300-345 `icnv{n}`): the input as
 * TWO tensors -- channels [0, c1) of the descriptor's Cin from x1 (N x H x W x
 * c1), the rest from x2 (N x H x W x (Cin - c1)) -- so that the concatenated
 * tensor is never written, and its data gradient into two tensors likewise.
 * c1 a multiple of 32 (forward, weight gradient) / of 64 when Cin is one (data
 * gradient: the kernel's block of output channels), else LSI_EINVAL.
 * lsi_conv2d_fwd_cat: bn_workspace != NULL adds the batch-norm sums as
 * lsi_conv2d_fwd_bnstats does (groups is ignored otherwise).
 * lsi_conv2d_wgrad_cat: x2 may be NULL (one tensor, c1 ignored);
 * weight_layout 0 writes g_weight contiguously (Cout x Cin x KH x KW), 2 with
 * torch's channels-last strides (Cout x KH x KW x Cin in memory: the layout of
 * the parameter after module.to(memory_format=torch.channels_last) -- autograd
 * then takes the gradient as it is instead of copying it into that layout).
 */
int lsi_conv2d_fwd_cat(const LsiConvDesc* d, const void* x1, const void* x2, int32_t c1,
                       const void* packed, void* out, float* bn_workspace, int32_t groups,
                       lsi_stream_t stream);
int lsi_conv2d_bwd_data_cat(const LsiConvDesc* d, const void* gy, const void* packed,
                            void* gx1, void* gx2, int32_t c1, lsi_stream_t stream);
int lsi_conv2d_wgrad_cat(const LsiConvDesc* d, const void* x1, const void* x2, int32_t c1,
                         const void* gy, float* g_weight, int32_t weight_layout,
                         void* workspace, size_t workspace_bytes, lsi_stream_t stream);
/*
 * All of the above behind one entry, with a workspace: lsi_conv2d_run(d, mode,
 * io) is lsi_conv2d_fwd (mode 0) / lsi_conv2d_bwd_data (mode 1) with the
 * options of the _bnstats and _cat variants taken from `io` -- and, when
 * `workspace` holds lsi_conv2d_workspace_bytes(d, mode) bytes (16-byte
 * aligned), the contraction SPLIT OVER THE INPUT CHANNELS for the launches
 * whose tiles alone do not fill the chip: the bottleneck maps of the U-Net
 * (reference nets.py:281-305, `cnv5` ... `icnv6`: 2 x 6 ... 16 x 48 pixels with
 * 256 - 1024 channels) are 64 - 256 tiles, each a chain of 16 - 32 dependent
 * stages; ks splits each multiply Cin / ks channels into fp32 partial sums in
 * the workspace, a second kernel adds them in a fixed order, rounds to bf16,
 * stores (into the two tensors of a skip connection's gradient, too) and takes
 * the batch-norm sums.  lsi_conv2d_workspace_bytes returns 0 where the kernel
 * does not split; a smaller or absent workspace is not an error (no split).
 * The caller owns the workspace (no state in the library); calls that share one
 * must be ordered on one stream.  Results are deterministic for a given
 * geometry and differ from the unsplit kernel's only by the fp32 summation
 * order.
 *   x, x2, c1 (mode 0): the input as one (x2 NULL) or two tensors;
 *   out, out2, c1 (mode 1): the gradient into one (out2 NULL) or two tensors;
 *   bn_workspace (NULL: none), groups: as lsi_conv2d_fwd_bnstats.
 */
typedef struct LsiConvIO {
  const void* x;
  const void* x2;
  const void* packed;
  void* out;
  void* out2;
  float* bn_workspace;
  void* workspace;
  size_t workspace_bytes;
  int32_t c1, groups;
} LsiConvIO;
size_t lsi_conv2d_workspace_bytes(const LsiConvDesc* d, int32_t mode);
int lsi_conv2d_run(const LsiConvDesc* d, int32_t mode, const LsiConvIO* io,
                   lsi_stream_t stream);
/*
 * Weight gradient of the convolution LsiConvDesc describes (TF autodiff of
 * slim.conv2d, reference nets.py:29-114, 244-348):
 *   g_weight[co][ci][ky][kx] = sum over n, oy, ox of
 *       gy[n][oy][ox][co] * x[n][oy*stride + ky - pad_t][ox*stride + kx - pad_l][ci]
 *   x: bf16 N x H x W x Cin, gy: bf16 N x OH x OW x Cout (16-byte aligned);
 *   g_weight: fp32 Cout x Cin x KH x KW, written (not added to).  MFMA with K =
 *   output pixels, both operands transposed by the LDS transpose read; partial
 *   sums of the pixel blocks in the workspace, folded by a second kernel
 *   (deterministic).  lsi_conv2d_wgrad_workspace_bytes returns 0 for shapes it
 *   does not take (LSI_EUNSUPPORTED from the call: channel counts that are not
 *   multiples of 32, or small maps with so many channels that the partial sums
 *   would exceed 96 MB -- the bottleneck layers, which stay on the library).
 *   For a transposed convolution described as above (its forward = the data
 *   gradient of the descriptor's convolution), x is the transposed
 *   convolution's OUTPUT gradient and gy its INPUT.
 */
size_t lsi_conv2d_wgrad_workspace_bytes(const LsiConvDesc* d);
int lsi_conv2d_wgrad(const LsiConvDesc* d, const void* x, const void* gy, float* g_weight,
                     void* workspace, size_t workspace_bytes, lsi_stream_t stream);

/*
 * The networks' FIRST convolution (reference nets.py:273 `cnv1`, :53 in
 * encoder_simple: slim.conv2d(inp_img, 32, [7, 7], stride=2), TF `SAME`, batch
 * norm + ReLU behind it) on the matrix cores: d->Cin <= 4 (the image's 3
 * channels), Cout = 32, 7 x 7, stride 2 -- anything else LSI_EUNSUPPORTED
 * (lsi_conv2d_first_supported tells beforehand).  A pixel is padded to four
 * channels in LDS and the K of one MFMA is a whole kernel row (7 taps x 4
 * channels); no packed weights: the kernels read the fp32 parameter itself
 * (weight_layout 0: contiguous Cout x Cin x KH x KW; 2: torch's channels-last
 * strides) and round to bf16 as torch.autocast does.
 *   x: the image, N x H x W x Cin with the channels innermost, fp32 (x_bf16 =
 *   0: rounded to bf16 on the way in) or bf16 (1).  out: bf16 N x OH x OW x 32.
 *   bn_workspace != NULL: the epilogue leaves the batch-norm sums of the rounded
 *   outputs for lsi_bn_relu_norm (as lsi_conv2d_fwd_bnstats; groups | N).
 * lsi_conv2d_first_wgrad: g_weight[co][c][ky][kx] = sum over n, oy, ox of
 *   gy[n][oy][ox][co] * x[n][2 oy + ky - pad_t][2 ox + kx - pad_l][c]  (written,
 *   fp32, in weight_layout; K = output pixels on the matrix cores, partial sums
 *   per workgroup in the workspace folded in a fixed order: deterministic).
 *   gy: bf16 N x OH x OW x 32, 16-byte aligned.  The image has no data gradient.
 */
int lsi_conv2d_first_supported(const LsiConvDesc* d);
int lsi_conv2d_first_fwd(const LsiConvDesc* d, const void* x, int32_t x_bf16,
                         const float* weight, int32_t weight_layout, void* out,
                         float* bn_workspace, int32_t groups, lsi_stream_t stream);
size_t lsi_conv2d_first_wgrad_workspace_bytes(const LsiConvDesc* d);
int lsi_conv2d_first_wgrad(const LsiConvDesc* d, const void* x, int32_t x_bf16, const void* gy,
                           float* g_weight, int32_t weight_layout, void* workspace,
                           size_t workspace_bytes, lsi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LSI_HIP_H_ */
