/*
 * gsseg.h -- C ABI of libgsseg_hip.so: the MI355X (gfx950) kernels under the
 * GenSeg segmentation hot path (U-Net / Pix2Pix / Dice+BCE forward + backward).
 *
 * The reference (importZL/semantic_segmentation) has no FFI of its own: its
 * boundary for this path is the Python object model.  Each entry
 * point below therefore cites the reference *operator call site* whose ATen
 * kernel(s) it replaces.  The Python package `semantic_segmentation_amd` binds
 * these with ctypes (semantic_segmentation_amd/_lib.py) and re-creates the
 * reference's module/function API on top.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch's caching
 *     allocator); kernels never allocate.  Workspaces are caller supplied.
 *   - activations / gradients: NHWC, 16-bit (GS_F16 or GS_BF16), with an explicit
 *     pixel stride (elements) and channel offset so a tensor can be a channel
 *     slice of a wider concat buffer.  Statistics, losses, weight gradients: fp32.
 *   - all launches go to `stream` (hipStream_t passed as void*), never synchronise.
 *   - return 0 on success, negative GsStatus otherwise; gs_last_error() gives text.
 */
#ifndef GSSEG_H
#define GSSEG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_ABI_VERSION 50

enum GsDtype { GS_F16 = 0, GS_BF16 = 1 };
enum GsStatus { GS_OK = 0, GS_EINVAL = -1, GS_ELAUNCH = -2, GS_EUNSUPPORTED = -3 };
enum GsAct { GS_ACT_NONE = 0, GS_ACT_RELU = 1, GS_ACT_LEAKY02 = 2, GS_ACT_TANH = 3 };

#define GS_MAX_TAPS 64

/* Geometry of one implicit-GEMM launch.  Logical output grid (N, OHg, OWg); logical pixel
 * (n, oy, ox) reads input pixel (oy*isy + tap_dy[t], ox*isx + tap_dx[t]) for tap t (zero when
 * outside [0,IH)x[0,IW)) and writes physical output pixel (oy*osy + ooy, ox*osx + oox) of an
 * [N, OH, OW] tensor.  This one form covers Conv2d (any k/stride/pad), its data gradient, and
 * ConvTranspose2d split into stride*stride sub-pixel classes. */
typedef struct GsConvGeom {
    int32_t N, IH, IW, Cin, in_pix_stride, in_coff;
    int32_t OHg, OWg, Cout;
    int32_t OH, OW, out_pix_stride, out_coff;
    int32_t isy, isx, osy, osx, ooy, oox;
    int32_t ntaps;
    int32_t tap_dy[GS_MAX_TAPS], tap_dx[GS_MAX_TAPS];
    int32_t tap_w[GS_MAX_TAPS];   /* weight slot of tap t: weights of tap t start at w + tap_w[t]*Cout*Cin (lets a
                                     sub-pixel class use a subset of a full [kh*kw][Cout][Cin] pack) */
    /* depth (Conv3d / ConvTranspose3d, GenSeg-3D/UNet3D/unet3d.py): a volume is a stack of images.  Logical image
     * index n' = nb*Dg + d; tap t reads input image nb*Din + d*isz + tap_dz[t] (zero outside [0,Din)) and the
     * output image is nb*Dout + d*osz + ooz.  2-D: Dg = Din = Dout = 1, isz = osz = 1, ooz = 0, tap_dz = 0. */
    int32_t Dg, Din, Dout, isz, osz, ooz;
    int32_t tap_dz[GS_MAX_TAPS];
} GsConvGeom;

const char* gs_last_error(void);
int gs_abi_version(void);

/* ---- MFMA implicit GEMM: nn.Conv2d / ConvTranspose2d forward and data-gradient -------------
 * replaces F.conv2d / F.conv_transpose2d at unet/unet_parts.py:16,19,53 and
 * models_pix2pix/networks.py:582,590-602,640-661 (and their autograd dgrad).
 * x: [N,IH,IW,*] dtype; w: [ntaps][Cout][Cin] dtype (K-major); y: [N,OH,OW,*] dtype.
 * bias (fp32 [Cout]) may be NULL.  bn_partials (fp32 [gs_conv_igemm_mtiles][2][Cout]) may be NULL;
 * when given, per-M-tile sums of y and y*y (fp32 accumulators, before rounding, before bias) are
 * written there for train-mode BatchNorm (unet_parts.py:17,20).  Requires Cin % 8 == 0. */
int gs_conv_igemm_mtiles(const GsConvGeom* g);
/* splitk_ws: optional split-K workspace.  Skinny GEMMs (few output tiles, long K: the 1x1 .. 16x16 levels of the Pix2Pix
 * generator at batch 2) are split over K, the parts store fp32 partial tiles in per-part slabs and the last part of a
 * tile sums them (in part order: deterministic) and runs the epilogue.  Caller-owned device memory, fp32,
 * gs_conv_igemm_workspace_floats() elements, zero-initialised ONCE (ticket counters return to zero); NULL = no split.
 * The library holds no state: launches sharing a workspace must be ordered on one stream (one workspace per
 * (device, stream) on the host side). */
int64_t gs_conv_igemm_workspace_floats(void);
int gs_conv_igemm(const GsConvGeom* g, const void* x, const void* w, void* y, const float* bias,
                  float* bn_partials, int act, int dtype, float* splitk_ws, int64_t splitk_ws_floats, void* stream);
/* Up to four such GEMMs over the same x / y / bias in ONE grid: the four sub-pixel classes of a stride-2 transposed
 * convolution (networks.py:486-511 merged kernel) or of the data gradient of a stride-2 convolution (networks.py:582,640-655).
 * g[i] / w[i] / bn_partials[i] (array may be NULL) per GEMM; <= 16 taps each, Cin != 8, one tile shape for all (equal
 * Cout and pixel counts).  GEMM i splits K through the i-th quarter of the workspace (a single gs_conv_igemm uses the
 * first quarter). */
int gs_conv_igemm_batch(int n, const GsConvGeom* const* g, const void* x, const void* const* w, void* y, const float* bias,
                        float* const* bn_partials, int act, int dtype, float* splitk_ws, int64_t splitk_ws_floats,
                        void* stream);

/* ---- stride-2 / kernel-2 transposed convolution as ONE pointwise GEMM + sub-pixel scatter ------
 * replaces nn.ConvTranspose2d(C, C/2, kernel_size=2, stride=2) at unet/unet_parts.py:51 (Up.up) and
 * nn.ConvTranspose3d(k=2, s=2) at GenSeg-3D/UNet3D/unet3d.py:68, including the zero F.pad offset of
 * unet_parts.py:58-61 (ooy/oox) and the write into the concat buffer of torch.cat (out_coff).
 * x: [N*D,IH,IW,*]; w: gs_pack_weight(transposed) slots [ncls][Cout][Cin], slot = (kz*2+ky)*2+kx, ncls = 4 (D == Dout
 * == 1) or 8; y: [N*Dout,OH,OW,*], voxel (2z+kz+ooz, 2y+ky+ooy, 2x+kx+oox).  Cout, strides, offsets % 8 == 0. */
int gs_upconv2x2_fwd(const void* x, const void* w, const float* bias, void* y, int N, int D, int IH, int IW, int Cin,
                     int in_pix_stride, int in_coff, int Cout, int Dout, int OH, int OW, int out_pix_stride,
                     int out_coff, int ooz, int ooy, int oox, int act, int dtype, void* stream);

/* Data gradient of the same ConvTranspose2d(kernel 2, stride 2) (unet/unet_parts.py:51,57; autograd of :57):
 * dx[n,iy,ix,ci] = sum_{a,b,co} dy[n, 2*iy+a+ooy, 2*ix+b+oox, co] * w[ci][co][a][b];  wd = the data-gradient pack [4][Cin][Cout]
 * of gs_pack_weight(transposed).  LDS-DMA GEMM over K = (sub-pixel class, co); covers power-of-two IH / IW, N*IH*IW % 256 == 0,
 * Cin % 128 == 0, Cout % 64 == 0, strides / offsets % 8 == 0 -- returns GS_EUNSUPPORTED for other shapes (no error string is
 * set; run gs_conv_igemm on the 4-tap stride-2 geometry instead). */
int gs_upconv2x2_dgrad(const void* dy, const void* wd, void* dx, int N, int IH, int IW, int Cin, int Cout, int OH, int OW,
                       int dy_pix_stride, int dy_coff, int ooy, int oox, int dx_pix_stride, int dx_coff, int dtype,
                       void* stream);

/* ---- 3x3 / stride 1 / pad 1 convolution with LDS halo reuse (the U-Net DoubleConv hot loop,
 * unet_parts.py:16,19) and its data gradient (pass flipped taps and the [9][Cin][Cout] pack).
 * Same operands as gs_conv_igemm; tap t reads input pixel (y + tap_dy[t], x + tap_dx[t]), offsets in [-1,1];
 * bn_partials: a buffer of gs_conv3x3_mtiles() rows of [2][Cout] fp32 always suffices; the launch WRITES
 * gs_conv3x3_stat_rows() rows -- one per block and cout-tile group on the LDS-DMA kernel (which keeps its sums in registers
 * across the items of a block), one per spatial patch on the others and in the pair forward -- and that is the row count
 * gs_bn_finalize / gs_bn_partials_colsum must be given.  (gs_conv3x3_stat_rows assumes the forward or flipped tap table and
 * channel strides / offsets that are multiples of 8; a launch with partials outside that fails with GS_EINVAL.) */
int gs_conv3x3_mtiles(int N, int H, int W, int Cout);
int gs_conv3x3_stat_rows(int N, int H, int W, int Cin, int Cout, int pair);
/* Diagnostics (process-wide, not part of the data path): which form of the kernel gs_conv3x3 launches where several apply:
 * -1 chosen by CU fill (default), 0 the register-staged big-K-step kernel, 4 / 8 the LDS-DMA kernel with that many waves per
 * block.  Every form computes the same sums in a form-specific order. */
int gs_conv3x3_set_kernel_form(int form);
/* Persistent-kernel grids (conv3x3, and the kernels that read gs_get_persistent_grid): at most `blocks` workgroups instead of
 * one per CU (256) -- data-parallel runs leave CUs to RCCL's kernels, which otherwise wait for a 152 KB-LDS block to retire.
 * 0 restores the default (GSSEG_C3_GRID or 256).  Changes gs_conv3x3_stat_rows(): set it before planning buffers. */
int gs_set_persistent_grid(int blocks);
int gs_get_persistent_grid(void);
int gs_conv3x3(const void* x, const void* w, void* y, const float* bias, float* bn_partials, int N, int H, int W,
               int Cin, int in_pix_stride, int in_coff, int Cout, int out_pix_stride, int out_coff,
               const int32_t* tap_dy, const int32_t* tap_dx, int act, int dtype, void* stream);

/* weight gradient of the same 3x3/s1/p1 convolution with halo reuse: dw fp32 [9][Cout][Cin] (ACCUMULATED with
 * atomics; caller zeroes) += sum_p dy[p][co] * x[p + tap][ci];  x / dy NHWC 16-bit with pixel stride / offset. */
int gs_conv3x3_wgrad(const void* x, const void* dy, float* dw, int N, int H, int W, int Cin, int in_pix_stride,
                     int in_coff, int Cout, int out_pix_stride, int out_coff, int dtype, void* stream);

/* ---- MFMA weight gradient -------------------------------------------------------------------
 * dw[t][co][ci] (fp32, ACCUMULATED with atomics: caller zeroes) += sum over logical pixels of
 * dy[pix][co] * x[inpix(pix,t)][ci].  `g` describes the forward conv: dy lives on the OUTPUT side
 * (N,OH,OW,out_pix_stride,out_coff), x on the INPUT side.  Replaces the autograd weight gradient of
 * the same call sites.  Requires Cin % 8 == 0 and Cout % 8 == 0. */
int gs_conv_wgrad(const GsConvGeom* g, const void* x, const void* dy, float* dw, int dtype, void* stream);
/* Same weight gradient WRITTEN instead of accumulated: dw[tap_w[t]][co][ci] = ... for the taps of g (other taps of dw
 * are not touched), so the caller need not zero dw.  Only valid when the launch does not split K
 * (gs_conv_wgrad_single_pass(g) == 1: the few-pixel layers of the batch-2 Pix2Pix step, where zero-filling and
 * read-modify-writing the fp32 gradient of a 512x1024x8x8 kernel costs more than computing it); otherwise GS_EINVAL. */
int gs_conv_wgrad_single_pass(const GsConvGeom* g);
int gs_conv_wgrad_assign(const GsConvGeom* g, const void* x, const void* dy, float* dw, int dtype, void* stream);
/* Deterministic form of gs_conv_wgrad: the gs_conv_wgrad_parts(g) K parts of the launch store their partial gradients in
 * slabs ws[part][tap][Cout][Cin] (gs_conv_wgrad_ws_floats(g) floats, no zero fill, no atomics); gs_wgrad_reduce_unpack
 * then sums the parts in order, scales and writes the reference layout.  Used for ConvTranspose2d(k2,s2) of `Up`
 * (unet/unet_parts.py:51): with it every gradient of the U-Net step is bit-reproducible. */
int gs_conv_wgrad_parts(const GsConvGeom* g);
int64_t gs_conv_wgrad_ws_floats(const GsConvGeom* g);
int gs_conv_wgrad_slabs(const GsConvGeom* g, const void* x, const void* dy, float* ws, int dtype, void* stream);
/* The same for up to four GEMMs over one x / dy in ONE grid (the sub-pixel classes of the merged transposed convolution,
 * networks.py:486-511): equal gradient sizes, tile shapes and K splits; GEMM i writes part p to
 * ws + p * (n * slab) + i * slab, slab = gs_conv_wgrad_ws_floats(g[i]) / gs_conv_wgrad_parts(g[i]) -- one slab of all n
 * gradients per part, so ONE gs_wgrad_reduce_unpack over [n*taps*Cout][Cin] sums every class; with a single part ws may be
 * the gradient tensor itself ([n][taps][Cout][Cin]). */
int gs_conv_wgrad_slabs_batch(int n, const GsConvGeom* const* g, const void* x, const void* dy, float* ws, int dtype,
                              void* stream);

/* ---- direct (VALU) convolutions for 1..4-channel ends of the nets ----------------------------
 * gs_conv_smallcin_fwd: x fp32 NCHW [N,Cin,IH,IW] (the image / mask as the loader hands it,
 *   unet_model.py:27, networks.py:582 outermost, :640) -> y NHWC dtype [N,OH,OW,Cout] (+bias,
 *   +act, +bn_partials as above; bn tile = 256 output pixels).  w fp32 [Cout][Cin][k][k] (reference layout).
 * gs_conv_smallcin_wgrad: dw fp32 [Cout][Cin][k][k] += gscale * sum dy*x (caller zeroes); deterministic two-stage
 *   reduction through the caller's workspace ws (gs_conv_direct_wgrad_ws_floats floats).
 * gs_conv_smallcin_dgrad: dx fp32 NCHW [N,Cin,IH,IW] = conv_transpose(dy, w)*gscale (overwrites). */
int gs_conv_smallcin_mtiles(int N, int OH, int OW);
int gs_conv_smallcin_fwd(const float* x, const float* w, const float* bias, void* y, float* bn_partials,
                         int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int k, int stride, int pad,
                         int act, int dtype, void* stream);
int64_t gs_conv_direct_wgrad_ws_floats(int N, int OH, int OW, int Cin, int Cout, int k);

/* ---- the one-channel stem WITHOUT its convolution output in memory ------------------------------------------------
 * (unet/unet_parts.py:16-18 with in_channels = 1: Conv2d(1, 64, 3, padding=1, bias=False) -> BatchNorm2d -> ReLU.)
 * The train-mode statistics of y = conv(x) are functions of the image and the 576 weights alone (sum_p y = sum_t w_t S_t,
 * sum_p y^2 = sum_tu w_t w_u G_tu with the tap sums S and the 9x9 tap Gram matrix G), so y need not exist:
 *   gs_stem_stats      tile partials [gs_conv_smallcin_mtiles(N,H,W)][2][64] for gs_bn_finalize, from x and w only;
 *   gs_stem_fwd_bn     z [N,H,W,64] 16-bit = act(conv(x) * scale + shift) in one pass (y stays in fp32 registers);
 *   gs_stem_bwd_onepass + gs_stem_bwd_finalize
 *                      the whole backward of the stage in ONE pass over z (the stored activation: only its sign, the
 *                      activation's mask, is used) and dz.  With g = dz*act', A[c][t] = sum_p g x_t(p), s1 = sum_p g:
 *                        sum_p g xhat = invstd_c (sum_t w[c][t] A[c][t] - mean_c s1_c)          (y = sum_t w_t x_t exactly)
 *                        dW[c][t] += gscale*scale_c*(A - c1_c S_t - c2_c invstd_c (sum_u w[c][u] G[u][t] - mean_c S_t))
 *                      with S / G the image's tap sums / tap Gram matrix, which gs_stem_stats also writes per tile
 *                      (tap_sums [mtiles][54], may be NULL there) and c1 = s1/count, c2 = sum g xhat / count (0 when
 *                      train_stats = 0: eval-mode statistics).  _onepass writes s1_partials [gs_stem_bwd_tiles][64] and the
 *                      A slabs ws [gs_stem_bwd_tiles][576]; _finalize (fp64) OVERWRITES dgamma / dbeta (gscale * sums; may be
 *                      NULL) and accumulates dw.
 * At batch 32, 256^2 this removes the write of y (268 MB), its read by gs_bn_act_apply and by the two backward passes.
 * x fp32 [N,1,H,W], w fp32 [64][1][3][3].  gs_stem_bwd_onepass returns GS_EUNSUPPORTED (no error string) when the image is
 * too wide for the LDS strip: the caller then re-forms y with gs_conv_smallcin_fwd and runs the tensor path. */
int gs_stem_stats(const float* x, const float* w, float* bn_partials, float* tap_sums, int N, int H, int W, void* stream);
int gs_stem_fwd_bn(const float* x, const float* w, const float* bn_scale, const float* bn_shift, int act, void* z, int N,
                   int H, int W, int dtype, void* stream);
int gs_stem_bwd_tiles(int N, int H, int W);
int gs_stem_bwd_onepass(const float* x, const void* z, const void* dz, int dz_stride, int dz_coff, int act,
                        float* s1_partials, float* ws, int N, int H, int W, int dtype, void* stream);
int gs_stem_bwd_finalize(const float* ws, const float* s1_partials, const float* tap_sums, const float* w,
                         const float* scale, const float* mean, const float* invstd, int train_stats, float gscale, float* dw,
                         float* dgamma, float* dbeta, int N, int H, int W, void* stream);

/* Stem backward in one pass (unet/unet_parts.py:16-18 with in_channels = 1, first stage of `inc`; the image needs no
 * gradient): BatchNorm(train)/activation backward apply fused with the weight gradient of the 1-channel 3x3/s1/p1
 * convolution with 64 outputs.  y [N,H,W,64] 16-bit = the convolution's output, dz = gradient w.r.t. the activation's
 * output ([N,H,W,*], pixel stride dz_stride, channel offset dz_coff), x fp32 [N,1,H,W]; scale/shift/mean/invstd from
 * gs_bn_finalize, c1/c2 from gs_bn_bwd_coeffs.  dw fp32 [64][1][3][3] += gscale * sum_pixels dy * x_tap with
 * dy = scale*(dz*act'(scale*y+shift) - c1 - xhat*c2) kept in fp32 registers -- the tensor dy is never materialised
 * (it replaces gs_bn_act_bwd_apply + gs_conv_smallcin_wgrad).  ws: gs_conv_direct_wgrad_ws_floats(N,H,W,1,64,3) floats.
 * Returns GS_EUNSUPPORTED (no error string) when the image is too wide for the LDS strip; the caller then runs the two
 * kernels it replaces. */
int gs_stem_bn_bwd_wgrad(const void* y, const void* dz, int dz_stride, int dz_coff, const float* x, const float* scale,
                         const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2,
                         int act, float* dw, float* ws, int N, int H, int W, float gscale, int dtype, void* stream);

/* workspace of the two below */
int gs_conv_smallcin_wgrad(const float* x, const void* dy, float* dw, float* ws, int N, int Cin, int IH, int IW,
                           int Cout, int OH, int OW, int k, int stride, int pad, float gscale, int dtype, void* stream);
int gs_conv_smallcin_dgrad(const void* dy, const float* w, float* dx, int N, int Cin, int IH, int IW, int Cout,
                           int OH, int OW, int k, int stride, int pad, float gscale, int dtype, void* stream);

/* ---- small-Cout head: 1x1 (OutConv, unet_parts.py:71-77) or kxk (PatchGAN last conv, networks.py:661)
 * x NHWC dtype [N,IH,IW,Cin] -> logits fp32 NCHW [N,Cout,OH,OW], Cout <= 4.  w fp32 [Cout][Cin][k][k].
 * bwd: dlogits fp32 NCHW (already multiplied by the loss scale) -> dx NHWC dtype; dw/db fp32 accumulate*gscale. */
int gs_conv_smallcout_fwd(const void* x, const float* w, const float* bias, float* y, int N, int IH, int IW, int Cin,
                          int Cout, int OH, int OW, int k, int stride, int pad, int dtype, void* stream);
int gs_conv_smallcout_bwd(const void* x, const float* w, const float* dy, void* dx, float* dw, float* db, float* ws,
                          int N, int IH, int IW, int Cin, int Cout, int OH, int OW, int k, int stride, int pad,
                          float gscale, int dtype, void* stream);

/* ---- train-mode BatchNorm2d split around the convolution (unet_parts.py:17,20; networks.py:584,586)
 * finalize: reduce [ntiles][2][C] partials -> scale = gamma*rsqrt(var+eps), shift = beta - mean*scale,
 *   save mean / invstd, update running_mean / running_var (momentum, unbiased var) in place when non-NULL.
 * eval mode: gs_bn_eval_coeffs builds scale/shift from the running statistics. */
/* floats a partials buffer must hold for `ntiles` tiles of C channels (includes the reduction scratch) */
int64_t gs_bn_partials_floats(int ntiles, int C);
int gs_bn_finalize(const float* partials, int ntiles, int C, double count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps,
                   float* scale, float* shift, float* mean, float* invstd, void* stream);
int gs_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* scale, float* shift, float* mean, float* invstd,
                      void* stream);

/* apply: z = act(y*scale + shift) (scale/shift NULL -> identity), y dense NHWC [N,H,W,C];
 * z written with (z_pix_stride, z_coff) -- e.g. straight into the skip half of the concat buffer
 * (unet_parts.py:67) ; optional 2x2/stride-2 max-pool of z (unet_parts.py:34) to zp [N,H/2,W/2,C] dense;
 * optional dropout keep-mask (uint8 [N,H,W,C], value scaled by 1/(1-p)) for networks.py:606-607. */
int gs_bn_act_apply(const void* y, const float* scale, const float* shift, int act, void* z, int z_pix_stride,
                    int z_coff, void* zp, const uint8_t* keep_mask, float keep_scale, int N, int H, int W, int C,
                    int dtype, void* stream);

/* backward of  z = act(bn(y)) [-> maxpool]:  dz = dz_a[pix*sa + coff_a + c] (may be NULL; optionally times a
 * dropout keep-mask, networks.py:606-607) + the max-pool gradient routed from dzp [N,H/2,W/2,C] (may be NULL;
 * first-max tie rule as ATen).  dz_b (dense, may be NULL) is a second consumer of the same pre-activation with
 * its own activation act_b: gh = dz*act'(v) + dz_b*act_b'(v)  (the generator's leaky-skip / relu-concat pair).
 * reduce: partial sums of dzh and dzh*xhat -> [gs_bn_bwd_tiles][2][C];
 * coeffs: from partials -> dgamma, dbeta (fp32, OVERWRITE, multiplied by gscale) and c1,c2 = sums/count;
 * apply:  dy = scale*(dzh - c1 - xhat*c2)   (bn==0: dy = dzh, act gradient only). */
int gs_bn_bwd_tiles(int N, int H, int W);                       /* upper bound, sizes the partials buffer */
int gs_bn_bwd_tiles_used(int N, int H, int W, int pooled);      /* tiles actually written -> gs_bn_bwd_coeffs */
int gs_bn_act_bwd_reduce(const void* y, const void* dz_a, int sa, int coff_a, const void* dzp,
                         const void* dz_b, int act_b, const uint8_t* keep_mask, float keep_scale,
                         const float* scale, const float* shift, const float* mean, const float* invstd, int act,
                         float* partials, int N, int H, int W, int C, int dtype, void* stream);
int gs_bn_bwd_coeffs(const float* partials, int ntiles, int C, double count, float gscale, float* dgamma,
                     float* dbeta, float* c1, float* c2, void* stream);
int gs_bn_act_bwd_apply(const void* y, const void* dz_a, int sa, int coff_a, const void* dzp,
                        const void* dz_b, int act_b, const uint8_t* keep_mask, float keep_scale,
                        const float* scale, const float* shift, const float* mean, const float* invstd,
                        const float* c1, const float* c2, int act, int bn, void* dy, int N, int H, int W, int C,
                        int dtype, void* stream);

/* MaxPool3d(2, stride 2) on NDHWC 16-bit volumes (GenSeg-3D/UNet3D/unet3d.py:37,44).  fwd: z [NB,D,H,W,*] (strided) ->
 * zp [NB,D/2,H/2,W/2,C].  bwd: dz (dense, OVERWRITE) = dres[pix*stride + coff + c] (skip gradient, may be NULL)
 * + dzp routed to the first maximum of each window (ATen scan order). */
int gs_maxpool3d_fwd(const void* z, int z_pix_stride, int z_coff, void* zp, int NB, int D, int H, int W, int C,
                     int dtype, void* stream);
/* pair forward (UNet3D): the maximum of the pair VALUES z_hi + z_lo over each window, stored as a pair (zp_lo may be NULL);
 * z_hi / z_lo point at the first channel of their planes inside a buffer of pixel stride z_pix_stride (unet3d.py:37,44). */
int gs_maxpool3d_fwd_pair(const void* z_hi, const void* z_lo, int z_pix_stride, void* zp_hi, void* zp_lo, int zp_pix_stride,
                          int NB, int D, int H, int W, int C, int dtype, void* stream);
/* ... with q planes (see gs_conv3x3_q8): z_q8 -> z_lo = byte 0 of the input buffer's q plane, the pooled channels start at its channel
 * zq_coff; zp_q8 -> zp_lo = byte 0 of the pooled buffer's q plane. */
int gs_maxpool3d_fwd_pair_q8(const void* z_hi, const void* z_lo, int z_q8, int zq_coff, int z_pix_stride, void* zp_hi, void* zp_lo,
                             int zp_q8, int zp_pix_stride, int NB, int D, int H, int W, int C, int dtype, void* stream);
int gs_maxpool3d_bwd(const void* z, int z_pix_stride, int z_coff, const void* dzp, const void* dres,
                     int res_pix_stride, int res_coff, void* dz, int NB, int D, int H, int W, int C, int dtype,
                     void* stream);

/* MaxPool2d(2) forward on its own (unet/unet_parts.py:34): z [N,H,W,*] (strided, e.g. the skip half of a concat buffer)
 * -> zp [N,H/2,W/2,C] dense (floor: an odd last row/column is dropped, as in ATen).  Used by the inference path, where
 * conv + folded BatchNorm + ReLU is ONE gs_conv3x3 call (bias + activation epilogue) and gs_bn_act_apply is not run. */
int gs_maxpool2x2_fwd(const void* z, int z_pix_stride, int z_coff, void* zp, int N, int H, int W, int C, int dtype,
                      void* stream);
/* The same on hi/lo pairs (folded-BatchNorm inference of the pair forward: conv + bias + ReLU leave the z pair, nn.MaxPool2d(2) of
 * `Down`, unet/unet_parts.py:34): the pooled pair = the maximum of the window's pair VALUES hi + lo, split again.  z_hi / z_lo: two
 * planes with pixel stride z_pix_stride (16-bit elements); zp_hi / zp_lo (zp_lo may be NULL) with zp_pix_stride. */
int gs_maxpool2x2_fwd_pair(const void* z_hi, const void* z_lo, int z_pix_stride, void* zp_hi, void* zp_lo, int zp_pix_stride, int N, int H,
                           int W, int C, int dtype, void* stream);

/* per-channel column sums over the sub-rectangle [y0,y0+h) x [x0,x0+w) of a (strided) NHWC tensor
 * [N,H,W,*]: out[c] (OVERWRITE) = gscale * sum t[pix*s + coff + c]  (bias gradient of ConvTranspose2d,
 * unet_parts.py:53, excluding the F.pad border of unet_parts.py:59-61).  ws: fp32 [1024*C] workspace. */
int gs_colsum(const void* t, int pix_stride, int coff, int N, int H, int W, int y0, int x0, int h, int w, int C,
              float gscale, float* ws, float* out, int dtype, void* stream);

/* The 1x1 head (OutConv, unet/unet_parts.py:74, 64 input channels, <= 4 classes) taking the last stage's convolution
 * OUTPUT y_conv [N,H,W,64] 16-bit plus that stage's BatchNorm scale / shift (gs_bn_finalize) and slope-family activation
 * (unet_parts.py:20-21): z = act(y_conv * scale + shift) is formed on the load path, in fp32 -- gs_bn_act_apply of the
 * last stage and the 2 x N*H*W*64*2 bytes of its output written and read back drop out.  _fwd: logits fp32 [N,Cout,H,W]
 * = z . w^T + bias.  _wgrad: dw [Cout][64] += gscale * sum_p dl[p] z[p], db += gscale * sum_p dl (ws as
 * gs_conv_smallcout_bwd: gs_conv_direct_wgrad_ws_floats(N,H,W,64,Cout,1)). */
int gs_head1x1_bn_fwd(const void* y_conv, const float* bn_scale, const float* bn_shift, int act, const float* w,
                      const float* bias, float* logits, int N, int H, int W, int Cout, int dtype, void* stream);
int gs_head1x1_bn_wgrad(const void* y_conv, const float* bn_scale, const float* bn_shift, int act, const float* w,
                        const float* dl, float* dw, float* db, float* ws, int N, int H, int W, int Cout, float gscale,
                        int dtype, void* stream);

/* BatchNorm + activation backward of the stage in FRONT of a pointwise head (unet/unet_parts.py:19-21 followed by the
 * OutConv of :74, n_classes <= 4): the gradient w.r.t. the activation's output is dz[p][c] = sum_k dl[n][k][h][w] *
 * w_head[k][c].  These two entry points take dl (fp32 [N,ncls,H,W], the logit gradient) and w_head (fp32 [ncls][C], the
 * head's weight) instead of a dz tensor and form dz on the fly in fp32 -- the head's data-gradient kernel
 * (gs_conv_smallcout_bwd with dx) and the 2 x N*H*W*C*2 bytes it writes and these passes read back drop out.  Same
 * partials / dy semantics as gs_bn_act_bwd_reduce / gs_bn_act_bwd_apply (bn = 1, plain pixels, no second source). */
int gs_bn_act_bwd_reduce_head(const void* y, const float* dl, const float* w_head, int ncls, const float* scale,
                              const float* shift, const float* mean, const float* invstd, int act, float* partials, int N,
                              int H, int W, int C, int dtype, void* stream);
int gs_bn_act_bwd_apply_head(const void* y, const float* dl, const float* w_head, int ncls, const float* scale,
                             const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2,
                             int act, void* dy, int N, int H, int W, int C, int dtype, void* stream);

/* The same bias gradient taken from the tile partials of the convolution that WROTE the tensor (gs_conv3x3 with
 * bn_partials, [ntiles][2][Cfull], sized by gs_bn_partials_floats): out[c] (OVERWRITE) = gscale * sum_tiles
 * partials[tile][0][coff + c].  For the un-padded case (the transposed convolution's output covers the whole skip
 * size, unet_parts.py:59-61 pads nothing): the data-gradient convolution's epilogue already holds the per-channel
 * sums in fp32, so the separate pass of gs_colsum over the tensor is not needed.  ntiles = gs_conv3x3_mtiles(...). */
int gs_bn_partials_colsum(const float* partials, int ntiles, int Cfull, int coff, int C, float gscale, float* out,
                          void* stream);

/* ---- weight packing ------------------------------------------------------------------------
 * Conv2d weight fp32 [Cout][Cin][kh][kw] -> fwd pack [kh*kw][Cout][Cin] and dgrad pack
 * [kh*kw][Cin][Cout] (tap order unchanged; the host flips taps through the geometry).
 * transposed=1: ConvTranspose2d weight fp32 [Cin][Cout][kh][kw], same outputs. Either output may be NULL.
 * unpack: dw fp32 [taps][A][B] -> grad fp32 [A][B][kh][kw] (transposed=0) or [B][A][kh][kw] (1), times gscale,
 * OVERWRITING grad. */
int gs_pack_weight(const float* w, void* w_fwd, void* w_dgrad, int Cout, int Cin, int taps, int transposed,
                   int dtype, void* stream);

/* The same packs for several conv weights in ONE launch (3x3 convs: taps = 9, k2/s2 transposed convs: taps = 4): a training
 * step re-packs every conv weight of the network after the optimiser step (unet/unet_model.py: 18 + 4 tensors); 22 launches of
 * a few microseconds each were launch-bound.  w_fwd or w_dgrad may be NULL. */
typedef struct GsPackDesc {
    const float* w;     /* fp32 parameter in the reference layout */
    void* w_fwd;        /* [taps][Cout][Cin] 16-bit, or NULL */
    void* w_dgrad;      /* [taps][Cin][Cout] 16-bit, or NULL */
    int32_t Cout, Cin, taps, transposed;
} GsPackDesc;
int gs_pack_weight_multi(int n, const GsPackDesc* descs, int dtype, void* stream);
int gs_unpack_wgrad(const float* dw, float* grad, int A, int B, int taps, int transposed, float gscale, void* stream);

/* Fake-image post-processing of the Unet step (running_files/train_end2end_jsrt.py:197-200): global min-max scaling to
 * [0,1] -> uint8 -> per-plane histogram equalisation (torchvision 0.14.1 F.equalize) -> gamma (F.adjust_gamma) -> float/255.
 * x, out: fp32 [N][hw], N = images x channels (one histogram per plane, min/max over everything); gamma_lut[256]: the
 * float the pipeline maps equalised level e to (uint8(255.999f * clamp((e/255)^gamma, 0, 1)) / 255, computed once by the
 * host); ws: gs_fake_postprocess_ws_floats(N) floats of scratch.  Bit-exact against oracle/postproc.py. */
int64_t gs_fake_postprocess_ws_floats(int N);
int gs_fake_postprocess(const float* x, float* out, float* ws, const float* gamma_lut, int N, int64_t hw, void* stream);

/* ISIC variant of the fake-image post-processing (running_files/train_end2end_isic.py:178-184,263-264): global min-max ->
 * uint8 -> [RandomEqualize] -> posterize(bits) -> [adjust_sharpness] -> [autocontrast] -> [adjust_saturation] -> float/255
 * (torchvision 0.14.1 algorithms).  The per-call random decisions are made by the host: *_on flags; blend ratios as the
 * pair (float(ratio), float(1.0 - ratio)) the reference's float arithmetic uses.  x, out: fp32 [N][C][H][W], C <= 4
 * (saturation needs C == 3); ws: gs_isic_fake_trans_ws_bytes(N*C, H*W) bytes.  Bit-exact against oracle/postproc.py. */
int64_t gs_isic_fake_trans_ws_bytes(int planes, int64_t hw);
int gs_isic_fake_trans(const float* x, float* out, void* ws, int N, int C, int H, int W, int equalize_on, int bits,
                       int sharpness_on, float sharp_r1, float sharp_r2, int autocontrast_on, int saturation_on, float sat_r1,
                       float sat_r2, void* stream);

/* Outermost generator layer (models_pix2pix/networks.py:588-593, merged 8x8 kernel): ConvTranspose2d(Cin -> Cout <= 4,
 * k 8, stride 2, pad 3) + bias + activation written as the fp32 NCHW image out [N,Cout,2h,2w]; u (may be NULL) receives
 * the pre-activation as 16-bit NHWC [N,2h,2w,cpad] for the backward pass.  x: 16-bit NHWC [N,h,w,*] (strided);
 * pack_fwd: the class-major pack [4][16][cpad][Cin] of gs_upconv_merge_pack (cpad == 8).  Direct VALU kernel: on the MFMA
 * engine this layer uses 1 of 64 N columns.  Needs (12*12*(Cin+8) + 64*Cout*Cin)*2 bytes <= 64 KB of LDS. */
int gs_upconv8_image_fwd(const void* x, int in_pix_stride, int in_coff, const void* pack_fwd, int cpad, const float* bias,
                         float* out, void* u, int N, int h, int w, int Cin, int Cout, int act, int dtype, void* stream);
/* Weight gradient of the same layer when it has ONE output channel (the JSRT generator, `output_nc = 1`): autograd of the merged
 * `MixedOp_upconv` transposed conv of the outermost `UnetSkipConnectionBlock` (models_pix2pix/networks.py:486-511,588-593) w.r.t. its
 * merged 8x8 kernel, written where gs_upconv_split_wgrad reads it: dwm[4 classes][16 taps][1][Cin] fp32 (un-scaled sums).
 * x = the layer's input [N,h,w,*] (16-bit, pixel stride x_pix_stride, Cin channels from channel 0), du = the gradient w.r.t. the
 * layer's pre-activation output [N,2h,2w,*] (16-bit, channel 0, pixel stride du_pix_stride).  Deterministic: per-block partials in
 * ws (gs_upconv8_image_wgrad_ws_floats floats, no initialisation needed) summed in block order.  gs_upconv8_image_wgrad_ok: Cout == 1,
 * Cin a multiple of 128 -- otherwise the caller takes gs_conv_wgrad_slabs_batch. */
int gs_upconv8_image_wgrad_ok(int Cin, int Cout);
int64_t gs_upconv8_image_wgrad_ws_floats(int N, int h, int w, int Cin);
int gs_upconv8_image_wgrad(const void* x, int x_pix_stride, const void* du, int du_pix_stride, float* ws, float* dwm, int N, int h, int w,
                           int Cin, int dtype, void* stream);

/* ---- Pix2Pix mixed transposed convolution (networks.py:486-511, operations.py:14-39) ------------------
 * The softmax-weighted sum of ConvTranspose2d k4p1 / k6p2 / k8p3 (stride 2) equals ONE k8/s2/p3 transposed
 * conv with Wm = s2*W8 + s1*pad1(W6) + s0*pad2(W4).  merge_pack builds, from the three fp32 parameters
 * [Cin][Cout][k][k] and the device vector softmax3 = softmax(arch[layer]), any of:
 *   pack_fwd   16-bit [4 classes][16 taps][Cout][Cin]  (class c = py*2+px, tap t = a*4+b, ky = 2a+1-py, kx = 2b+1-px;
 *              input offset of the tap: dy = 1-a+py, dx = 1-b+px)
 *   pack_dgrad 16-bit [64][Cin][Cout]                  (dX = stride-2, pad-3 conv of dY with taps (ky-3, kx-3))
 *   merged_f32 fp32 [Cin][Cout][8][8]
 * split_wgrad: dwm fp32 [4][16][Cout][Cin] -> dw4/dw6/dw8 (OVERWRITE, times gscale*softmax3[j]) and
 *   dots3[j] += gscale * <dwm restricted to kernel j's window, Wj>  (= dLoss/d softmax3[j]; caller zeroes). */
int gs_upconv_merge_pack(const float* w4, const float* w6, const float* w8, const float* softmax3, void* pack_fwd,
                         void* pack_dgrad, float* merged_f32, int Cin, int Cout, int dtype, void* stream);
int gs_upconv_split_wgrad(const float* dwm, const float* w4, const float* w6, const float* w8, const float* softmax3,
                          float gscale, float* dw4, float* dw6, float* dw8, float* dots3, int Cin, int Cout,
                          void* stream);
/* Same split with the three architecture dot products summed in a fixed order: every block stores its partial sums in ws
 * (gs_upconv_split_wgrad_ws_floats(Cin, Cout) floats) and one small kernel adds them to dots3 in block order. */
int64_t gs_upconv_split_wgrad_ws_floats(int Cin, int Cout);
int gs_upconv_split_wgrad_det(const float* dwm, const float* w4, const float* w6, const float* w8, const float* softmax3,
                              float gscale, float* dw4, float* dw6, float* dw8, float* dots3, float* ws, int Cin, int Cout,
                              void* stream);
/* The same split reading `nparts` split-K slabs of the merged weight gradient ([64 slots][Cout][Cin] each, `part_stride` floats
 * apart: what gs_conv_wgrad_slabs_batch leaves for the four sub-pixel classes) and summing them in part order on the way --
 * instead of gs_wgrad_reduce_unpack writing the summed 134 MB (1024 -> 512) and this pass reading them back.
 * gs_upconv_split_wgrad_parts_ok: 1 when the shape is covered (Cin % 32 == 0, Cout % 8 == 0). */
int gs_upconv_split_wgrad_parts_ok(int Cin, int Cout);
int gs_upconv_split_wgrad_parts(const float* slabs, int nparts, int64_t part_stride, const float* w4, const float* w6,
                                const float* w8, const float* softmax3, float gscale, float* dw4, float* dw6, float* dw8,
                                float* dots3, float* ws, int Cin, int Cout, void* stream);

/* layout helpers: fp32 NCHW <-> 16-bit NHWC */
int gs_nchw_to_nhwc(const float* src, void* dst, int N, int C, int H, int W, int dst_pix_stride, int dst_coff,
                    int dtype, void* stream);
int gs_nhwc_to_nchw(const void* src, int src_pix_stride, int src_coff, float* dst, int N, int C, int H, int W,
                    float gscale, int dtype, void* stream);

/* ---- losses (train_end2end_jsrt.py:136-138,181-183; util/dice_score.py:5-28; networks.py:263-281)
 * seg loss forward: logits fp32 NCHW [N,C,H,W], mask uint8 [N,H,W] (class index; {0,1} for C==1).
 *   C==1: BCEWithLogits(mean) + 1 - dice(sigmoid(x), t) with ONE global sum over the batch;
 *   C>1 : CrossEntropy(mean) + 1 - dice(softmax(x), onehot(t)) (global sum over N*C*H*W).
 *   out[0]=loss out[1]=ce/bce out[2]=dice_loss out[3..5]=inter(2*sum p t), sum p, sum t, out[6]=1 (multiplier of the
 *   Dice gradient: data-parallel callers all-reduce out[3..5], rewrite out[0], out[2] and set out[6]=world for the exact
 *   global-batch Dice of dice_score.py:10).  out has 8 floats.  ws: fp32 [4*1024].
 * backward: dlogits fp32 NCHW = dloss/dlogits * gscale * gout[0]  (gout: device scalar, upstream grad). */
int gs_seg_loss_fwd(const float* logits, const uint8_t* mask, int N, int C, int H, int W, float* ws, float* out,
                    void* stream);
int gs_seg_loss_bwd(const float* logits, const uint8_t* mask, const float* out, const float* gout, float gscale,
                    float* dlogits, int N, int C, int H, int W, void* stream);
/* generic dice_loss(input, target) on fp32 tensors of n elements, one global sum (dice_score.py:25-28 with
 * reduce_batch_first=True): out[0]=loss, out[1..3]=inter,sum_p,sum_t.  bwd: dinput = dloss/dinput * gout[0]. */
/* Per-item Dice coefficients in ONE launch pair (util/dice_score.py:5-17 with reduce_batch_first=False, the form
 * unet/evaluate.py:29-43 averages): item b = n_per consecutive elements of p / t; out[0] = mean over the B items,
 * out[1 + b] = dice_b.  ws: gs_dice_batched_ws_floats(B) floats.
 * gs_eval_dice fuses the prediction of unet/evaluate.py:31,38-41 in front of it: sigmoid(logit) > 0.5 (C == 1) or the
 * arg-max class (first maximum), one item per (sample, foreground class); logits fp32 NCHW, mask uint8 [N][HW];
 * ws for B = N * max(1, C - 1) items, out[1 + N*max(1,C-1)]. */
int64_t gs_dice_batched_ws_floats(int B);
int gs_dice_coeff_batched(const float* p, const float* t, int B, int64_t n_per, float* ws, float* out, void* stream);
int gs_eval_dice(const float* logits, const uint8_t* mask, int N, int C, int64_t HW, float* ws, float* out, void* stream);
/* the ISIC script's validation metric (running_files/train_end2end_isic.py:58-84): sigmoid(logit) > 0.5, per-sample Jaccard
 * index (I + 1) / (P + T - I + 1), mean over the batch; one class; ws / out as gs_eval_dice with B = N. */
int gs_eval_jaccard(const float* logits, const uint8_t* mask, int N, int64_t HW, float* ws, float* out, void* stream);
/* BCE + Jaccard loss of the ISIC variant (running_files/train_end2end_isic.py:40-56,247-249; one class): per sample
 * jac_i = (sum p t + 1) / (sum (p + t) - sum p t + 1), loss = mean BCEWithLogits + 1 - mean_i jac_i.  logits fp32 [N][HW],
 * mask uint8 [N][HW]; out: gs_jaccard_loss_out_floats(N) floats = {loss, bce, 1 - mean jac, 0, (I_i, S_i) per sample} (the
 * backward pass reads it); ws: gs_dice_batched_ws_floats(N).  bwd: dlogits (OVERWRITE) = gout * gscale * dloss/dlogits. */
int64_t gs_jaccard_loss_out_floats(int N);
int gs_jaccard_seg_loss_fwd(const float* logits, const uint8_t* mask, int N, int64_t HW, float* ws, float* out, void* stream);
int gs_jaccard_seg_loss_bwd(const float* logits, const uint8_t* mask, const float* out, const float* gout, float gscale,
                            float* dlogits, int N, int64_t HW, void* stream);
int gs_dice_loss_fwd(const float* p, const float* t, int64_t n, float* ws, float* out, void* stream);
int gs_dice_loss_bwd(const float* t, const float* out, const float* gout, float* dp, int64_t n, void* stream);
/* mean-reduced elementwise losses: mode 0 = BCEWithLogits vs constant label `cval` (GANLoss vanilla),
 * 1 = MSE vs constant (lsgan), 2 = mean(x)*cval (wgangp, cval=+-1), 3 = L1 |x - t| (t tensor),
 * 4 = BCEWithLogits vs tensor t.   out[0] = loss.  bwd: dx = dloss/dx * gout[0] * gscale. */
int gs_mean_loss_fwd(const float* x, const float* t, float cval, int mode, int64_t n, float* ws, float* out,
                     void* stream);
int gs_mean_loss_bwd(const float* x, const float* t, float cval, int mode, int64_t n, const float* gout,
                     float gscale, float* dx, void* stream);

/* ---- precise mode of the U-Net forward (opt-in) ---------------------------------------------
 * The 16-bit engine's logits differ from the fp32 reference (unet/unet_model.py:26-37) by up to ~4e-3 (23 stacked 16-bit
 * roundings); in precise mode activations and weights travel as PAIRS of 16-bit values v = hi + lo (hi = 16-bit(v),
 * lo = 16-bit(v - hi)) and the MFMA contractions run over the K concatenation [x_hi | x_lo | x_hi] . [w_hi | w_hi | w_lo]
 * (exact products, one fp32 accumulator), which brings the logits to ~1e-5 of the reference.
 * gs_pack_weight_split: w fp32 [Cout][Cin][taps] (transposed: [Cin][Cout][taps]) -> pack [taps][Cout][3*Cin] 16-bit.
 * gs_conv3x3_precise: as gs_conv3x3 with x = [hi | lo] planes (in_wrap channels per pixel; K chunk c reads input chunk
 *   c mod in_wrap), w = split pack with K = 3*Cin, result as the pair y_hi / y_lo (same stride / offset).
 * gs_upconv2x2_fwd_precise: the same for Up.up (unet_parts.py:51).
 * gs_conv_smallcin_fwd_split: first conv (fp32 image, fp32 weights) -> y pair; bn_partials as gs_conv_smallcin_fwd.
 * gs_bn_act_apply_split: z = act(scale * (y_hi + y_lo) + shift) -> z pair (both with z_pix_stride / z_coff), optional 2x2
 *   max-pooled pair (pixel stride zp_pix_stride).   gs_head1x1_fwd_split: OutConv on a dense pair, Cin == 64. */
int gs_pack_weight_split(const float* w, void* pack, int Cout, int Cin, int taps, int transposed, int dtype, void* stream);
/* Mixed-precision plans (UNet(precise="mixed"): the pair forward with the MFMA segments chosen per stage): a general segment
 * pack.  pack[t][co][k], k over the concatenation of nseg <= GS_SEG_MAX segments; segment j = input channels
 * [ci0[j], ci0[j] + len[j]) of hi(w) (kind 0), lo(w) (kind 1) or zeros (kind 2: K padding).  [hi | hi | lo] over all channels reproduces
 * gs_pack_weight_split; one launch packs all n descriptors. */
#define GS_SEG_MAX 4
typedef struct GsSegPackDesc {
    const float* w;     /* fp32 parameter in the reference layout ([Cout][Cin][taps]; transposed: [Cin][Cout][taps]) */
    void* pack;         /* [taps][Cout][sum of len] 16-bit */
    int32_t Cout, Cin, taps, transposed, nseg;
    int32_t kind[GS_SEG_MAX], ci0[GS_SEG_MAX], len[GS_SEG_MAX];
} GsSegPackDesc;
int gs_pack_weight_segs(int n, const GsSegPackDesc* descs, int dtype, void* stream);
int gs_conv3x3_precise(const void* x, const void* w, void* y_hi, void* y_lo, const float* bias, float* bn_partials, int N,
                       int H, int W, int K, int in_pix_stride, int in_coff, int in_wrap, int Cout, int out_pix_stride,
                       int out_coff, const int32_t* tap_dy, const int32_t* tap_dx, int act, int dtype, void* stream);
/* ---- "q" stages: the correction terms of a pair-forward conv as ONE FP8 block-scaled MFMA segment ----------
 * A conv on hi/lo pairs needs x_hi.w_hi + x_lo.w_hi + x_hi.w_lo to meet 1e-3 on the logits (unet_parts.py:16,19 are plain fp32); the
 * two correction terms are 2^-11 of the main product, so 4 significant bits suffice for them: they run as e4m3 operands on
 * v_mfma_scale_f32_32x32x64_f8f6f4 (twice the 16-bit MFMA rate) -- a "q" stage costs 2x the MFMA work of the plain conv, not 3x.
 * Q PLANE: per 32 channels one 64-byte chunk [lo8: 32 x e4m3(lo * 2^(XH+LS)) | hi8: 32 x e4m3(hi * 2^XH)], XH = -2, LS = 11 (fp16);
 *   it takes the place of the 16-bit lo plane ([hi plane | q plane] = the same bytes per pixel).
 * gs_pack_weight_q8: fp32 conv weight [Cout][Cin][taps] -> pack[t][co] = [w_hi 16-bit (Cin) | per 32 channels: w_hi8 (32) | w_lo8 (32)]
 *   (4*Cin bytes) and wexp[co] = the power-of-two exponent shared by the row's e4m3 planes (from the row's amax); n descriptors, 1 launch.
 * gs_conv3x3_q8 / gs_conv3d_3x3x3_q8: the conv (forward taps) -> dense pair y_hi / y_lo + BatchNorm partials as gs_conv3x3_precise;
 *   LDS-DMA kernel only: gs_conv3x3_q8_ok(W, Cin, Cout) (W >= 24, Cin %% 64 == 0, Cout %% 8 == 0), fp16 only.
 * gs_bn_act_apply_split_q8: gs_bn_act_apply_split writing q planes instead of 16-bit lo planes (z_q8 / zp_q8: which outputs).
 * gs_stem_fwd_bn_pair_q8: gs_stem_fwd_bn_pair with a q plane.   gs_q8_from_hi: the q plane (lo8 = 0) of channels stored hi-only. */
typedef struct GsQ8PackDesc {
    const float* w;     /* fp32 parameter [Cout][Cin][taps] */
    void* pack;         /* [taps][Cout][4*Cin bytes] */
    int32_t* wexp;      /* [Cout] */
    int32_t Cout, Cin, taps;
} GsQ8PackDesc;
int gs_pack_weight_q8(int n, const GsQ8PackDesc* descs, int dtype, void* stream);
int gs_conv3x3_q8_ok(int W, int Cin, int Cout);
int gs_conv3x3_q8(const void* x, const void* w, const int32_t* wexp, void* y_hi, void* y_lo, float* bn_partials, int N, int H, int W,
                  int Cin, int in_pix_stride, int in_coff, int Cout, int out_pix_stride, int out_coff, int dtype, void* stream);
int gs_conv3d_3x3x3_q8(const void* x, const void* w, const int32_t* wexp, void* y_hi, void* y_lo, float* bn_partials, int NB, int D,
                       int H, int W, int Cin, int in_pix_stride, int in_coff, int Cout, int out_pix_stride, int out_coff, int dtype,
                       void* stream);
int gs_bn_act_apply_split_q8(const void* y_hi, const void* y_lo, const float* scale, const float* shift, int act, void* z_hi,
                             void* z_lo, int z_q8, int z_pix_stride, int z_coff, void* zp_hi, void* zp_lo, int zp_q8,
                             int zp_pix_stride, int N, int H, int W, int C, int dtype, void* stream);
int gs_stem_fwd_bn_pair_q8(const float* x, const float* w, const float* bn_scale, const float* bn_shift, int act, void* z_hi,
                           void* z_q, int z_pix_stride, int N, int H, int W, int dtype, void* stream);
int gs_q8_from_hi(const void* x, void* q, int64_t pixels, int C, int pix_stride, int coff, int dtype, void* stream);
/* gs_conv3d_3x3x3_precise: Conv3d(k3, p1) of the pair forward of UNet3D (GenSeg-3D/UNet3D/unet3d.py:28-31,69-71; BASELINE config 5):
 *   gs_conv3d_3x3x3 with the K extent / in_wrap / [27][Cout][K] segment pack / y pair of gs_conv3x3_precise. */
int gs_conv3d_3x3x3_precise(const void* x, const void* w, void* y_hi, void* y_lo, const float* bias, float* bn_partials,
                            int NB, int D, int H, int W, int K, int in_pix_stride, int in_coff, int in_wrap, int Cout,
                            int out_pix_stride, int out_coff, const int32_t* tap_dz, const int32_t* tap_dy,
                            const int32_t* tap_dx, int act, int dtype, void* stream);
/* gs_conv3d_3x3x3_precise with the wrapped part of K continuing at input channel in_wrap_to (a multiple of 64; in_wrap_to + K - in_wrap <=
 * in_wrap) instead of channel 0: the decoder-entry convs of UNet3D (unet3d.py:80-81, input torch.cat((up, residual), 1) stored as
 * [up_h res_h | res_l]) run their w_lo segment over the residual channels only. */
int gs_conv3d_3x3x3_precise_to(const void* x, const void* w, void* y_hi, void* y_lo, const float* bias, float* bn_partials, int NB, int D,
                               int H, int W, int K, int in_pix_stride, int in_coff, int in_wrap, int in_wrap_to, int Cout, int out_pix_stride,
                               int out_coff, const int32_t* tap_dz, const int32_t* tap_dy, const int32_t* tap_dx, int act, int dtype,
                               void* stream);
int gs_upconv2x2_fwd_precise(const void* x, const void* w, const float* bias, void* y_hi, void* y_lo, int N, int IH, int IW,
                             int K, int in_pix_stride, int in_coff, int in_wrap, int Cout, int OH, int OW,
                             int out_pix_stride, int out_coff, int ooy, int oox, int dtype, void* stream);
int gs_conv_smallcin_fwd_split(const float* x, const float* w, void* y_hi, void* y_lo, float* bn_partials, int N, int Cin,
                               int H, int W, int Cout, int k, int pad, int dtype, void* stream);
int gs_bn_act_apply_split(const void* y_hi, const void* y_lo, const float* scale, const float* shift, int act, void* z_hi,
                          void* z_lo, int z_pix_stride, int z_coff, void* zp_hi, void* zp_lo, int zp_pix_stride, int N,
                          int H, int W, int C, int dtype, void* stream);
/* The same with MaxPool3d(2) (GenSeg-3D/UNet3D/unet3d.py:29-36: BatchNorm3d + ReLU + pooling of an analysis block): the z pair of the
 * [NB*D, H, W] voxel grid (even D, H, W) AND the pooled pair [NB*D/2, H/2, W/2] (maximum of the stored pair values) in one pass --
 * bit-identical to gs_bn_act_apply_split + gs_maxpool3d_fwd_pair without reading the z pair back.  z_lo / zp_lo may be NULL. */
int gs_bn_act_apply_split_pool3d(const void* y_hi, const void* y_lo, const float* scale, const float* shift, int act, void* z_hi, void* z_lo,
                                 int z_pix_stride, int z_coff, void* zp_hi, void* zp_lo, int zp_pix_stride, int NB, int D, int H, int W, int C,
                                 int dtype, void* stream);
int gs_head1x1_fwd_split(const void* x_hi, const void* x_lo, const float* w, const float* bias, float* y, int N, int H,
                         int W, int Cin, int Cout, int dtype, void* stream);
/* Pair-forward forms of the "never stored" edge kernels (mixed mode; gs_stem_fwd_bn / gs_head1x1_bn_fwd of the default engine):
 * gs_head1x1_bn_fwd_split: OutConv on act(scale * (y_hi + y_lo) + shift) of the last stage's conv-output pair.
 * gs_stem_fwd_bn_pair: the one-channel stem conv + BatchNorm + activation in one pass -> z_hi / z_lo (pixel stride
 *   z_pix_stride, z_lo may be NULL).   gs_stem_bwd_onepass_strided: gs_stem_bwd_onepass with z at a pixel stride.
 * gs_bn_act_apply_split accepts z_lo == NULL / zp_lo == NULL when no consumer reads the lo plane. */
int gs_head1x1_bn_fwd_split(const void* y_hi, const void* y_lo, const float* scale, const float* shift, int act, const float* w,
                            const float* bias, float* logits, int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
int gs_stem_fwd_bn_pair(const float* x, const float* w, const float* bn_scale, const float* bn_shift, int act, void* z_hi,
                        void* z_lo, int z_pix_stride, int N, int H, int W, int dtype, void* stream);
int gs_stem_bwd_onepass_strided(const float* x, const void* z, int z_pix_stride, const void* dz, int dz_stride, int dz_coff,
                                int act, float* s1_partials, float* ws, int N, int H, int W, int dtype, void* stream);

/* Weight gradient of ConvTranspose2d(kernel 2, stride 2) (unet/unet_parts.py:51,57) as a pointwise GEMM with K = pixels and
 * LDS-DMA operands (csrc/upwgrad.hip): x = the layer's input [N][IH][IW][x_pix_stride], dy = the gradient of its output
 * inside [N][OH][OW][dy_pix_stride] (output pixel (2y + a + ooy, 2x + b + oox)).  gs_upconv2x2_wgrad_parts() slabs
 * [4][Cin][Cout] (class = 2a + b) go to ws (gs_upconv2x2_wgrad_ws_floats() floats) and are summed in order by
 * gs_wgrad_reduce_unpack(ws, parts, grad, Cin, Cout, 4, 0, gscale) into the reference layout [Cin][Cout][2][2].  parts == 0 /
 * GS_EUNSUPPORTED: shape outside the kernel (power-of-two maps, IW >= 16, Cin % 128 == 0, Cout % 64 == 0): use gs_conv_wgrad_slabs. */
int gs_upconv2x2_wgrad_parts(int N, int IH, int IW, int Cin, int Cout);
int64_t gs_upconv2x2_wgrad_ws_floats(int N, int IH, int IW, int Cin, int Cout);
int gs_upconv2x2_wgrad_slabs(const void* x, const void* dy, float* ws, int N, int IH, int IW, int Cin, int x_pix_stride,
                             int x_coff, int Cout, int OH, int OW, int dy_pix_stride, int dy_coff, int ooy, int oox, int dtype,
                             void* stream);

/* Deterministic weight gradient of the 3x3 conv without atomics: gs_conv3x3_wgrad_slabs stores every split-K part's
 * tile into its own slab ws[part][9][Cout][Cin] (fp32, gs_conv3x3_wgrad_ws_floats() elements, no zero fill needed,
 * gs_conv3x3_wgrad_parts() parts); gs_wgrad_reduce_unpack sums the parts in order, scales by gscale and writes the
 * gradient in the reference layout ([A][B][taps]; [B][A][taps] when transposed) -- it replaces the zero fill, the
 * atomics of gs_conv3x3_wgrad and the gs_unpack_wgrad pass (autograd of unet_parts.py:16,19). */
int64_t gs_conv3x3_wgrad_ws_floats(int N, int H, int W, int Cin, int Cout);
int gs_conv3x3_wgrad_parts(int N, int H, int W, int Cin, int Cout);
int gs_conv3x3_wgrad_slabs(const void* x, const void* dy, float* ws, int N, int H, int W, int Cin, int in_pix_stride,
                           int in_coff, int Cout, int out_pix_stride, int out_coff, int dtype, void* stream);
int gs_wgrad_reduce_unpack(const float* ws, int nparts, float* grad, int A, int B, int taps, int transposed,
                           float gscale, void* stream);

/* ---- 3x3x3 / stride 1 / pad 1 Conv3d on the halo-reuse kernels ------------------------------------
 * replaces nn.Conv3d(kernel_size=3, padding=1) forward / data gradient / weight gradient of
 * GenSeg-3D/UNet3D/unet3d.py:28-31 (Conv3DBlock) and :69-71 (UpConv3DBlock).  Volumes are NB*D depth slices
 * [NB*D, H, W, *]; w / dw: [27][Cout][Cin], slot = kd*9 + kh*3 + kw.  tap_dz[kd] / tap_dy,tap_dx[kh*3+kw] are the
 * input offsets each slot reads (forward: kd-1, kh-1, kw-1; data gradient: the negated offsets with the [27][Cin][Cout]
 * pack).  bn_partials: [gs_conv3d_3x3x3_mtiles][2][Cout].  Requires Cin % 8 == 0, Cout % 8 == 0, 16-byte aligned
 * output channel slices (other shapes: gs_conv_igemm with depth taps). */
int gs_conv3d_3x3x3_mtiles(int NB, int D, int H, int W, int Cout);
int gs_conv3d_3x3x3(const void* x, const void* w, void* y, const float* bias, float* bn_partials, int NB, int D, int H,
                    int W, int Cin, int in_pix_stride, int in_coff, int Cout, int out_pix_stride, int out_coff,
                    const int32_t* tap_dz, const int32_t* tap_dy, const int32_t* tap_dx, int act, int dtype, void* stream);
int gs_conv3d_3x3x3_wgrad(const void* x, const void* dy, float* dw, int NB, int D, int H, int W, int Cin,
                          int in_pix_stride, int in_coff, int Cout, int out_pix_stride, int out_coff, int dtype,
                          void* stream);
/* deterministic form (slabs ws[part][27][Cout][Cin], then gs_wgrad_reduce_unpack with taps = 27) */
int64_t gs_conv3d_3x3x3_wgrad_ws_floats(int NB, int D, int H, int W, int Cin, int Cout);
int gs_conv3d_3x3x3_wgrad_parts(int NB, int D, int H, int W, int Cin, int Cout);
int gs_conv3d_3x3x3_wgrad_slabs(const void* x, const void* dy, float* ws, int NB, int D, int H, int W, int Cin,
                                int in_pix_stride, int in_coff, int Cout, int out_pix_stride, int out_coff, int dtype,
                                void* stream);

/* ---- bilinear x2 up-sampling, align_corners=True -----------------------------------------------
 * replaces nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) + F.pad + torch.cat of the bilinear=True
 * U-Net (unet/unet_parts.py:49-50,58-67) and its autograd.  x: [N,IH,IW,*] 16-bit NHWC (pixel stride / channel
 * offset); y: the [N,OH,OW,*] concat buffer, the 2IH x 2IW result lands at (ooy, oox) in channels out_coff...
 * _bwd reads the same slice of the concat-buffer gradient and writes d(x) (gather form, deterministic). */
int gs_upsample2x_bilinear_fwd(const void* x, void* y, int N, int IH, int IW, int C, int in_pix_stride, int in_coff,
                               int OH, int OW, int out_pix_stride, int out_coff, int ooy, int oox, int dtype, void* stream);
/* the same interpolation on a hi/lo pair (pair forward of the bilinear=True U-Net, unet_parts.py:49-50): the planes of x share
 * (in_pix_stride, in_coff), those of y (out_pix_stride, out_coff); y_lo may be NULL. */
int gs_upsample2x_bilinear_fwd_pair(const void* x_hi, const void* x_lo, void* y_hi, void* y_lo, int N, int IH, int IW, int C,
                                    int in_pix_stride, int in_coff, int OH, int OW, int out_pix_stride, int out_coff, int ooy,
                                    int oox, int dtype, void* stream);
int gs_upsample2x_bilinear_bwd(const void* dy, void* dx, int N, int IH, int IW, int C, int dy_pix_stride, int dy_coff,
                               int OH, int OW, int dx_pix_stride, int dx_coff, int ooy, int oox, int dtype, void* stream);

/* ---- on-device mask augmentation: one affine warp per sample ----------------------------------------
 * replaces the host-side imgaug pipeline of running_files/train_end2end_jsrt.py:99-112 applied at :186-190 (Fliplr,
 * CropAndPad, Affine scale / translate / rotate / shear in random order -- every stage is an affine map, their
 * composition is one) and the re-binarisation of :191-193.  src/dst: fp32 [N,C,H,W]; mats: [N][6] row-major 2x3,
 * mapping DESTINATION pixel centres to source coordinates: (sx, sy) = M (x+.5, y+.5, 1) - .5; bilinear taps, zeros
 * outside; thresh >= 0: dst = (value > thresh) ? 1 : 0, thresh < 0: the interpolated value. */
int gs_affine_warp(const float* src, float* dst, const float* mats, int N, int C, int H, int W, float thresh,
                   void* stream);

/* ---- multi-tensor optimiser steps (one launch for a whole model) ---------------------------------
 * replaces optim.RMSprop(net.parameters(), lr, weight_decay=1e-8, momentum=0.9, foreach=True).step()
 * (running_files/train_end2end_jsrt.py:69-70) and torch.optim.Adam(...).step() (models_pix2pix/pix2pix_model.py:69-72,
 * train_end2end_jsrt.py:318), with torch's single-tensor arithmetic (torch/optim/rmsprop.py, adam.py; not centered,
 * not amsgrad, L2 weight decay).  All tables live in device memory: per tensor the pointers and element count, per chunk
 * of gs_optim_chunk_elems() elements the owning tensor and the chunk start.  grad_scale multiplies every gradient first
 * (1/world_size, or a loss-scale inverse).  momentum_buf entries may be NULL when momentum == 0.
 * gs_optim_adam: step_scalars[2t] = lr / (1 - beta1^step_t), step_scalars[2t+1] = sqrt(1 - beta2^step_t). */
int gs_optim_chunk_elems(void);
int gs_optim_rmsprop(float* const* params, const float* const* grads, float* const* square_avg,
                     float* const* momentum_buf, const int64_t* sizes, const int32_t* chunk_tensor,
                     const int64_t* chunk_start, int nchunks, float lr, float alpha, float eps, float weight_decay,
                     float momentum, float grad_scale, void* stream);
int gs_optim_adam(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                  const int64_t* sizes, const int32_t* chunk_tensor, const int64_t* chunk_start, int nchunks,
                  const float* step_scalars, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSSEG_H */
