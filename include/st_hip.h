/*
 * st_hip.h -- C ABI of libst_hip.so: hand-written CDNA4 (gfx950 / MI355X) kernels for
 * the AdaIN style-transfer hot path of vTuanpham/Style_transfer.
 *
 * The reference has no FFI of its own: the hot path sits behind torch.nn.Module.__call__
 * on contiguous NCHW fp32 tensors.  Each entry point below replaces the
 * stock ATen op sequence of the reference lines cited next to it.  The host mirror in
 * style_transfer_amd/ (same class names / signatures as the reference) binds these with
 * ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous NCHW float32 unless stated;
 *   - the caller owns all memory (inputs, outputs, workspaces); the library never
 *     allocates, frees or keeps a pointer past the call;
 *   - kernels are enqueued asynchronously on `stream` (a hipStream_t; 0 = null stream);
 *   - return 0 on success, a negative ST_ERR_* otherwise; st_last_error() gives a
 *     thread-local message.  Nothing aborts or throws across the ABI.
 */
#ifndef ST_HIP_H
#define ST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* st_stream_t; /* hipStream_t */

#define ST_OK 0
#define ST_ERR_BAD_ARG (-1)   /* null pointer, non-positive or unsupported shape */
#define ST_ERR_LAUNCH (-2)    /* hipGetLastError() after the launch was not hipSuccess */
#define ST_ERR_UNSUPPORTED (-3)

#define ST_PAD_REPLICATE 0 /* SymmetricPadding2D((1,1,1,1)) == edge replicate, generator.py:14-39 */
#define ST_PAD_ZERO 1      /* torchvision VGG19 Conv2d(padding=1), trainer.py:168 */

#define ST_POOL_FLOOR 0 /* torchvision vgg19 MaxPool2d(2,2), trainer.py:168 */
#define ST_POOL_CEIL 1  /* MaxPool2d((2,2),(2,2),(0,0),ceil_mode=True), generator.py:53,60,73 */

#define ST_HIST_INVERSE_QUADRATIC 0 /* losses.py:228-234 */
#define ST_HIST_RBF 1               /* losses.py:221-227 */
#define ST_HIST_THRESHOLDING 2      /* losses.py:218-220 */

/* ---- library ------------------------------------------------------------------------ */
int st_version(void);
const char* st_last_error(void);

/* ---- 3x3 convolution (MFMA implicit GEMM) ---------------------------------------------
 * Replaces  SymmetricPadding2D((1,1,1,1)) -> nn.Conv2d(cin,cout,(3,3)) [-> nn.ReLU()]
 * (generator.py:47-76 encoder, :89-117 decoder), optionally preceded by
 * nn.UpsamplingNearest2d(scale_factor=2) (generator.py:92,105,112), and the zero-padded
 * Conv2d(3x3,padding=1)+ReLU(inplace) of the torchvision VGG19 loss network
 * (trainer.py:165-199).
 *
 * Weights are consumed in a packed K-major layout produced by st_conv3x3_pack_weights
 * from the reference's OIHW nn.Parameter (the packed copy is a caller-owned cache).
 */

/* number of floats of the packed weight buffer for a cout x cin x 3 x 3 filter */
size_t st_conv3x3_packed_elems(int cout, int cin);

/* OIHW [cout][cin][3][3] -> packed.  transpose_flip = 0: forward filter.
 * transpose_flip = 1: the data-gradient filter (in/out channels swapped, taps rotated
 * 180 degrees); then the packed buffer describes a cin-out x cout-in filter and must be
 * sized with st_conv3x3_packed_elems(cin, cout). */
int st_conv3x3_pack_weights(const float* w_oihw, float* w_packed, int cout, int cin,
                            int transpose_flip, st_stream_t stream);

/* y[n][cout][H][W] = act(bias + conv3x3(pad(up(x))))   with  H = (upsample2x?2:1)*hin.
 * x: [n][cin][hin][win].  bias may be NULL.  relu: 0/1.  pad_mode: ST_PAD_*. */
int st_conv3x3_fwd(const float* x, const float* w_packed, const float* bias, float* y,
                   int n, int cin, int hin, int win, int cout,
                   int upsample2x, int pad_mode, int relu, st_stream_t stream);

/* The same layer followed by MaxPool2d((2,2),(2,2),(0,0),ceil_mode=True) (generator.py:51-53, 58-60,
 * 71-73) in ONE launch: ypool[n][cout][ceil(hin/2)][ceil(win/2)]; the full-resolution activation is
 * never written.  Forward only (the encoder is frozen, trainer.py:286-297). */
int st_conv3x3_pool_fwd(const float* x, const float* w_packed, const float* bias, float* ypool,
                        int n, int cin, int hin, int win, int cout, int pad_mode, int relu,
                        st_stream_t stream);

/* Sub-pixel form of the forward pass of a layer that reads its input through the nearest x2 upsample
 * (generator.py:92,105,112 -> :93,106,113; replicate padding): output pixel (2i+a, 2j+b) only sees 2 x 2 source
 * pixels, with the filter rows / columns that land on the same source pixel summed when the weights are packed --
 * four 2 x 2 convolutions over the LOW-resolution input, 16 instead of 36 multiply-adds per source pixel and channel
 * pair.  Same result as st_conv3x3_fwd(upsample2x = 1, ST_PAD_REPLICATE) up to fp32 summation order (the summed
 * weights).  x [n][cin][hin][win] -> y [n][cout][2 hin][2 win].  Packed buffer: st_conv3x3_up2_packed_elems() floats. */
size_t st_conv3x3_up2_packed_elems(int cout, int cin);
int st_conv3x3_pack_weights_up2(const float* w_oihw, float* w_packed, int cout, int cin, st_stream_t stream);
int st_conv3x3_up2_fwd(const float* x, const float* w_packed_up2, const float* bias, float* y, int n, int cin,
                       int hin, int win, int cout, int relu, st_stream_t stream);

/* Data gradient of the layer above (decoder backward, trainer.py:472; loss-VGG backward).
 *   dy:      [n][cout][h][w]   gradient w.r.t. the layer output (h, w = OUTPUT dims)
 *   y_gate:  nullable; the layer's post-ReLU output: dy is taken as 0 where y_gate <= 0
 *   w_packed_t: st_conv3x3_pack_weights(..., transpose_flip = 1)
 *   dx:      [n][cin][h/s][w/s], s = 2 if the layer had the fused upsample, else 1
 *   dx_gate: nullable, same shape as dx (ST_PAD_REPLICATE only): dx is written as 0 where
 *            dx_gate <= 0 -- the ReLU backward of the layer that PRODUCED this layer's input, fused
 *            here so that layer's own dgrad / wgrad run without a y_gate (one tensor read less each)
 * ST_PAD_ZERO: one MFMA launch writes dx.  ST_PAD_REPLICATE: the gradient is evaluated on the
 * padded (h+2) x (w+2) domain into `workspace` (st_conv3x3_dgrad_workspace_bytes) and folded
 * back onto the replicated edges (and 2x2-summed for the upsample) by a second launch. */
size_t st_conv3x3_dgrad_workspace_bytes(int n, int cin, int h, int w, int pad_mode);
int st_conv3x3_dgrad(const float* dy, const float* y_gate, const float* w_packed_t, float* dx,
                     float* workspace, const float* dx_gate, int n, int cout, int h, int w, int cin,
                     int upsample2x, int pad_mode, st_stream_t stream);

/* Weight and bias gradient (decoder training, trainer.py:286-291, 471-473).
 *   x: [n][cin][hin][win] layer input; dy / y_gate: [n][cout][H][W], H = (upsample2x?2:1)*hin
 *   dw: [cout][cin][3][3] (OIHW, overwritten); dbias: [cout] or NULL
 * Split-K partial slabs go to `workspace` (st_conv3x3_wgrad_workspace_bytes) and are summed in
 * a fixed order: deterministic, no float atomics.  dy and y_gate must be 16-byte aligned. */
size_t st_conv3x3_wgrad_workspace_bytes(int n, int cout, int cin, int h, int w);
int st_conv3x3_wgrad(const float* x, const float* dy, const float* y_gate, float* dw,
                     float* dbias, float* workspace, int n, int cin, int hin, int win, int cout,
                     int upsample2x, int pad_mode, st_stream_t stream);

/* ---- bf16 compute mode of the same layers (BASELINE config 4: bf16 compute, fp32 master) ------
 * Same layer semantics, shapes and fp32 NCHW activations as st_conv3x3_fwd / st_conv3x3_dgrad
 * above (generator.py:47-76, 89-117; trainer.py:165-199); only the MFMA operands are bf16:
 * activations are rounded to nearest-even while staged into LDS, weights when packed, products
 * accumulate in fp32 (v_mfma_f32_32x32x16_bf16).  The packed buffer holds
 * st_conv3x3_packed_elems_bf16() 2-byte elements and must be 16-byte aligned. */
size_t st_conv3x3_packed_elems_bf16(int cout, int cin);
int st_conv3x3_pack_weights_bf16(const float* w_oihw, void* w_packed, int cout, int cin,
                                 int transpose_flip, st_stream_t stream);
int st_conv3x3_fwd_bf16(const float* x, const void* w_packed, const float* bias, float* y,
                        int n, int cin, int hin, int win, int cout,
                        int upsample2x, int pad_mode, int relu, st_stream_t stream);
int st_conv3x3_pool_fwd_bf16(const float* x, const void* w_packed, const float* bias, float* ypool,
                             int n, int cin, int hin, int win, int cout, int pad_mode, int relu,
                             st_stream_t stream);
/* bf16 mode, layer-to-layer form: the activation on either side may be in the channel-group layout "G8" =
 * bf16 [n][c/8][h][w][8] (x_g8 / y_g8 != 0; 0 = fp32 NCHW as above).  The conv -> conv hand-overs inside one
 * Sequential (generator.py:47-76, 89-117) then move 2 instead of 4 bytes per value, a G8 input is staged by
 * LDS-DMA, and the results are bit-identical to chaining st_conv3x3_fwd_bf16 / st_conv3x3_pool_fwd_bf16 (a G8
 * output is the round-to-nearest-even the next layer's staging applies anyway).  pool != 0: fused ceil-mode 2x2
 * max pool (not with upsample2x).  A G8 input needs cin % 16 == 0, a G8 output cout % 8 == 0; G8 buffers must be
 * 16-byte aligned.  Both padding modes (replicate: generator.py:19-39; zero: the loss VGG, trainer.py:165-199).
 * Forward only. */
int st_conv3x3_fwd_bf16_g8(const void* x, int x_g8, const void* w_packed, const float* bias, void* y, int y_g8,
                           int n, int cin, int hin, int win, int cout, int upsample2x, int pad_mode, int relu,
                           int pool, st_stream_t stream);
/* bf16 mode, data gradient of a ZERO-padded layer inside a G8 chain: the backward of the loss VGG
 * (loss.backward() of trainer.py:472 through the extractors of trainer.py:165-199, 446-457).  The same implicit GEMM
 * on w_packed_t (st_conv3x3_pack_weights_bf16 with transpose_flip), no bias, no activation.  dy [n][cout][h][w] is
 * fp32 NCHW or (dy_g8) G8; a G8 dy must ALREADY carry this layer's ReLU gate (its producer applied it: dx_gate_g8 of
 * the launch before, or st_grad_join_g8).  dx [n][cin][h][w] is fp32 NCHW or (dx_g8) G8; dx_gate_g8 (G8, the shape
 * of dx, needs dx_g8; may be NULL) = this layer's INPUT activation: dx is stored as 0 where it is <= 0, i.e. it
 * leaves the launch gated for the layer below.  Bit-identical to st_conv3x3_dgrad_bf16 on the same values (the G8
 * rounding is the one the next launch's staging applies to an fp32 gradient). */
int st_conv3x3_dgrad_bf16_g8(const void* dy, int dy_g8, const void* w_packed_t, void* dx, int dx_g8,
                             const void* dx_gate_g8, int n, int cout, int h, int w, int cin, st_stream_t stream);
/* bf16 mode, data gradient of a REPLICATE-padded decoder layer (generator.py:85-130; loss.backward() of
 * trainer.py:472) inside a G8 chain.  dy [n][cout][h][w]: fp32 NCHW or (dy_g8) G8, already gated.  The result --
 * the adjoint of the edge-replicate pad (generator.py:19-39) and of the nearest x2 upsample applied to the
 * padded-domain data gradient, on the source pixels [n][cin][h >> up][w >> up], zeroed where dx_gate (the layer's
 * input activation, fp32 NCHW or (dx_gate_g8) G8, may be NULL) is <= 0 -- is written in every form requested (each
 * may be NULL): dx_f32 fp32 NCHW; dx_g8 G8 (operand of the next st_conv3x3_dgrad_bf16_g8_chain); dx_g16 bf16
 * [n][cin][h][w] and bias_slabs [st_conv3x3_dgrad_chain_bias_slabs(n, cin, h, w, up)][cin] fp32 partial sums
 * (operands of st_conv3x3_wgrad_bf16_pre of the layer below; every row is written).  Two forms, same fp32 values per
 * element: (1) with more than 32 channels, an even (x4 under the upsample) h, w and a source grid of at least 4 x 4: the zero-padded conv on the UNPADDED grid with fold / gate / hand-over forms
 * in its epilogue, then two small launches that redo the border source pixels from the padded-domain values of the
 * ring (csrc/dgrad_band.hip; `workspace` holds those); (2) otherwise the padded-domain result goes to `workspace`
 * and a second pass folds it.  `workspace`: st_conv3x3_dgrad_workspace_bytes(n, cin, h, w, ST_PAD_REPLICATE) for
 * both.  Needs cin % 8 == 0 and (w >> up) % 4 == 0.  Same fp32 values as st_conv3x3_dgrad_bf16 with dx_gate. */
int st_conv3x3_dgrad_bf16_g8_chain(const void* dy, int dy_g8, const void* w_packed_t, float* workspace,
                                   const void* dx_gate, int dx_gate_g8, float* dx_f32, void* dx_g8, void* dx_g16,
                                   float* bias_slabs, int n, int cout, int h, int w, int cin, int upsample2x,
                                   st_stream_t stream);
int st_conv3x3_dgrad_chain_bias_slabs(int n, int cin, int h, int w, int upsample2x);
/* rows per image written by form (2) alone (kept for callers of the round-2 interface) */
int st_conv3x3_dgrad_fold8_bias_slabs(int h, int w, int upsample2x);
/* Weight gradient of a decoder layer whose gated output gradient arrives READY from the launch above: g16 bf16
 * [n][cout][H][W] and bias_slabs [n_bias_slabs][cout] (summed in a fixed order into dbias; dbias may be NULL); x is
 * the layer input, fp32 NCHW or (x_g8) G8 [n][cin/8][hin][win][8].  Same GEMM as st_conv3x3_wgrad_bf16's staged form
 * (needs W % 32 == 0, H % 4 == 0, cout != 3 where H, W = the conv-input dims after the upsample; the workspace size
 * is 0 for other shapes: use st_conv3x3_wgrad_bf16 there). */
size_t st_conv3x3_wgrad_bf16_pre_workspace_bytes(int n, int cout, int cin, int h, int w);
int st_conv3x3_wgrad_bf16_pre(const void* x, int x_g8, const void* g16, const float* bias_slabs, int n_bias_slabs,
                              float* dw, float* dbias, float* workspace, int n, int cin, int hin, int win, int cout,
                              int upsample2x, int pad_mode, st_stream_t stream);
/* Batched square fp32 products on the fp32 matrix pipe: c[b] = alpha * op(a[b]) . op(b[b]) + diag * I, n x n
 * row-major matrices, n % 64 == 0, op = transpose where trans_a / trans_b.  The C x C products of
 * VincentStyleLoss.l2wass_dist (losses.py:67-81: V sqrt(L) V^T, root_cov_stl @ cov_synth @ root_cov_stl and their
 * gradients, V diag(g) V^T behind eigvalsh) and of the Newton-Schulz square root that stands in for eigh(cov_stl);
 * replaces torch.matmul / torch.bmm / torch.baddbmm (rocBLAS) there.  c may not alias a or b. */
int st_bmm_f32(const float* a, const float* b, float* c, int batch, int n, int trans_a, int trans_b, float alpha,
               float diag, st_stream_t stream);
/* Weight / bias gradient of the decoder's 3-output-channel top layer (generator.py:127-128; loss.backward() of
 * trainer.py:472) when its input arrives in G8 [n][cin/8][hin][win][8] from the bf16 chain: st_conv3x3_wgrad's
 * vector-ALU form for cout == 3 (fp32 dy [n][3][H][W], fp32 accumulation, fixed-order slab sums) reading the bf16
 * activation directly -- the values the layer's forward multiplied.  cin % 8 == 0; workspace:
 * st_conv3x3_wgrad_workspace_bytes(n, 3, cin, H, W); dbias may be NULL. */
int st_conv3x3_wgrad_cout3_xg8(const void* x_g8, const float* dy, float* dw, float* dbias, float* workspace, int n,
                               int cin, int hin, int win, int upsample2x, int pad_mode, st_stream_t stream);
/* The gradient that reaches a CAPTURED layer output (a loss tap of trainer.py:446-457, fp32 NCHW) or the input of
 * a max pool, formed in one pass as the G8 operand of that layer's st_conv3x3_dgrad_bf16_g8:
 *     out_g8 = [x_gate > 0] * (chain + add_post) + add_pre
 * chain: gradient from the layer above, fp32 [n][c][h][w]; or chain_pooled: fp32 [n][c][h/2][w/2], the gradient
 * w.r.t. the floor-mode 2x2 max pool of x_gate (torchvision vgg19 'M'), routed like st_maxpool2x2_bwd (h, w even);
 * add_post / add_pre: loss gradients w.r.t. the post- / pre-ReLU capture (IntermediateLayerGetter returns the last
 * requested index before its in-place ReLU, trainer.py:165-199); x_gate: the captured tensor (NULL: no gate).
 * Every input is optional (at least one source); c % 8 == 0.  Same values as st_maxpool2x2_bwd / fp32 add /
 * st_relu_bwd followed by the bf16 rounding of the data-gradient kernel's staging. */
int st_grad_join_g8(const float* chain, const float* chain_pooled, const float* x_gate, const float* add_post,
                    const float* add_pre, void* out_g8, int n, int c, int h, int w, st_stream_t stream);
/* The same with a loss gradient handed over as per-plane AFFINE COEFFICIENTS instead of a tensor: the gradient of
 * AdaINStyleLoss (losses.py:14-39) w.r.t. a tap t is k0 + k1 * (t - mean) per (image, channel) plane, so the join
 * evaluates it from the tap it reads anyway and the [n][c][h][w] fp32 gradient is never written or re-read.
 * aff_post / aff_pre: [n * c][4] = (k0, k1, mean, -) from st_meanstd_loss_bwd_coeffs, in place of add_post / add_pre
 * (either may be NULL; not together with the tensor form of the same term); aff_x: the tap tensor (== x_gate when
 * that is given with chain_pooled).  Same fp32 expression as st_meanstd_loss_bwd: identical values. */
int st_grad_join_g8_affine(const float* chain, const float* chain_pooled, const float* x_gate, const float* add_post,
                           const float* add_pre, const float* aff_x, const float* aff_post, const float* aff_pre,
                           void* out_g8, int n, int c, int h, int w, st_stream_t stream);
/* The same for the bf16x3 mode (weights from st_conv3x3_pack_weights_bf16x3): a "G8" activation is the hi/lo pair
 * form bf16 [n][c/8][2][h][w][8] with part 0 = bf16(v) and part 1 = bf16(v - part 0) -- the split the kernels apply
 * to an fp32 activation, done once by the producing layer; as many bytes as fp32, staged by LDS-DMA.  Bit-identical
 * to chaining st_conv3x3_fwd_bf16x3 / st_conv3x3_pool_fwd_bf16x3. */
int st_conv3x3_fwd_bf16x3_g8(const void* x, int x_g8, const void* w_packed, const float* bias, void* y, int y_g8,
                             int n, int cin, int hin, int win, int cout, int upsample2x, int pad_mode, int relu,
                             int pool, st_stream_t stream);
/* workspace: st_conv3x3_dgrad_workspace_bytes() (same as the fp32 entry point) */
int st_conv3x3_dgrad_bf16(const float* dy, const float* y_gate, const void* w_packed_t, float* dx,
                          float* workspace, const float* dx_gate, int n, int cout, int h, int w, int cin,
                          int upsample2x, int pad_mode, st_stream_t stream);

/* ---- "bf16x3": fp32-grade results from the bf16 matrix pipe ------------------------------------
 * Same layers, shapes, fp32 NCHW activations and fp32 OIHW master weights as st_conv3x3_fwd /
 * st_conv3x3_pool_fwd / st_conv3x3_dgrad (generator.py:47-76, 89-117; trainer.py:165-199).  Every fp32
 * operand v is split exactly into hi = bf16(v), lo = bf16(v - hi) (activations while staged into LDS,
 * weights when packed) and each product is evaluated as hi*hi + hi*lo + lo*hi with fp32 accumulation:
 * three v_mfma_f32_32x32x16_bf16 per product at 16x the fp32 MFMA rate, relative error of a product
 * < 2^-15 (measured on the 19-conv stylise pipeline: ~1e-5 of the fp32 result; the task's fp32
 * tolerance is 1e-3).  The packed buffer holds st_conv3x3_packed_elems_bf16x3() 2-byte elements
 * (hi and lo images), 16-byte aligned. */
size_t st_conv3x3_packed_elems_bf16x3(int cout, int cin);
int st_conv3x3_pack_weights_bf16x3(const float* w_oihw, void* w_packed, int cout, int cin,
                                   int transpose_flip, st_stream_t stream);
int st_conv3x3_fwd_bf16x3(const float* x, const void* w_packed, const float* bias, float* y,
                          int n, int cin, int hin, int win, int cout,
                          int upsample2x, int pad_mode, int relu, st_stream_t stream);
int st_conv3x3_pool_fwd_bf16x3(const float* x, const void* w_packed, const float* bias, float* ypool,
                               int n, int cin, int hin, int win, int cout, int pad_mode, int relu,
                               st_stream_t stream);
int st_conv3x3_dgrad_bf16x3(const float* dy, const float* y_gate, const void* w_packed_t, float* dx,
                            float* workspace, const float* dx_gate, int n, int cout, int h, int w, int cin,
                            int upsample2x, int pad_mode, st_stream_t stream);

/* bf16-operand weight / bias gradient (decoder training in the bf16 compute mode, generator.py:85-130 trained by
 * trainer.py:286-291, 471-473); same arguments and outputs (fp32 dw / dbias) as st_conv3x3_wgrad.  Its workspace is
 * st_conv3x3_wgrad_bf16_workspace_bytes(n, cout, cin, h_out, w_out) bytes (h_out / w_out: the OUTPUT size, i.e. twice
 * hin / win with upsample2x): the split-K slabs plus, for layers made of whole 4 x 32 pixel tiles, the two operands
 * in the form the GEMM consumes -- bf16(dy * gate) and the padded (upsampled) bf16 input -- written once per call. */
size_t st_conv3x3_wgrad_bf16_workspace_bytes(int n, int cout, int cin, int h, int w);
int st_conv3x3_wgrad_bf16(const float* x, const float* dy, const float* y_gate, float* dw,
                          float* dbias, float* workspace, int n, int cin, int hin, int win, int cout,
                          int upsample2x, int pad_mode, st_stream_t stream);

/* ---- SymmetricPadding2D.forward, any padding (generator.py:14-39) ----------------------
 * numpy-"symmetric" reflection about the half-pixel points -0.5 and n-0.5 (edge pixel
 * repeated).  x: [nc][h][w] -> y: [nc][h+top+bottom][w+left+right].  The fused 3x3 conv
 * never materialises this tensor; the entry point exists for the stand-alone module. */
int st_sympad_fwd(const float* x, float* y, int nc, int h, int w, int left, int right, int top,
                  int bottom, st_stream_t stream);

/* ---- 1x1 convolution 3->3 on the raw image (generator.py:46) ------------------------- */
int st_conv1x1_3to3_fwd(const float* x, const float* w /*[3][3]*/, const float* b /*[3]*/,
                        float* y, int n, int hw, st_stream_t stream);

/* ---- transforms.Normalize (trainer.py:299, 440-442, 707-708): y = (x - mean[c]) / std[c].
 * mean_host / std_host: HOST arrays of c <= 8 floats (mean_host NULL = 0, which makes this the
 * backward as well: dx = dy / std). */
int st_normalize_fwd(const float* x, float* y, int n, int c, int hw, const float* mean_host,
                     const float* std_host, st_stream_t stream);

/* ---- stand-alone nn.ReLU (torchvision vgg19 ReLU(inplace=True)); the conv kernels fuse it, this
 * exists for the one place a pre-ReLU feature has to survive (shared loss-VGG pass).
 * bwd: dx = dy where y > 0 (y = the ReLU output). */
int st_relu_fwd(const float* x, float* y, long long n, st_stream_t stream);
int st_relu_bwd(const float* y, const float* dy, float* dx, long long n, st_stream_t stream);

/* ---- 2x2 stride-2 max pool (generator.py:53,60,73; torchvision vgg19 'M') ------------- */
int st_maxpool2x2_fwd(const float* x, float* y, int nc, int h, int w, int ceil_mode,
                      st_stream_t stream);

/* floor-mode pool of fp32 NCHW [n][c][h][w] written in G8 bf16 [n][c/8][h/2][w/2][8] (the input form of the next
 * conv of a bf16-mode chain, st_conv3x3_fwd_bf16_g8); c % 8 == 0, h and w even (floor == ceil).  Same bits as
 * st_maxpool2x2_fwd followed by the bf16 rounding of that conv's staging. */
int st_maxpool2x2_fwd_g8(const float* x, void* y_g8, int n, int c, int h, int w, st_stream_t stream);

/* The pool-input layers of a bf16-mode chain UNDER AUTOGRAD (the loss VGG's conv1_2 / conv2_2 / conv3_4 in front of
 * torchvision's 'M' layers, trainer.py:165-199, on the decoded image -- trainer.py:446-457, backward :472):
 * conv + ReLU + 2x2 max pool with the pooled activation in G8 [n][cout/8][h/2][w/2][8] and `codes`, one byte per
 * pooled pixel and channel in the same order ([n][cout/8][h/2][w/2][8] uint8): bits 0-1 = the window position
 * (0 1 / 2 3) of the maximum under torch.max_pool2d's backward rule (the first maximum in scan order), bit 2 = that
 * maximum is > 0.  The full-resolution activation is not written.  Even h and w; cout % 8 == 0; x as in
 * st_conv3x3_fwd_bf16_g8.  The pooled values are the bits st_conv3x3_fwd_bf16_g8(pool = 1) writes. */
int st_conv3x3_pool_fwd_bf16_g8_codes(const void* x, int x_g8, const void* w_packed, const float* bias,
                                      void* y_pooled_g8, void* codes, int n, int cin, int hin, int win, int cout,
                                      int pad_mode, int relu, st_stream_t stream);
/* A loss tap that feeds a pool (torchvision vgg19 conv1_2 / conv2_2 under IntermediateLayerGetter, trainer.py:165-199,
 * 446-457): y = the fp32 NCHW activation the loss reads, y_pooled_g8 = its 2x2 max pool in G8 [n][cout/8][h/2][w/2][8]
 * for the next conv -- one launch instead of st_conv3x3_fwd_bf16_g8 + st_maxpool2x2_fwd_g8 (same bits).  x is G8
 * (cin % 16 == 0); even h and w; cout % 8 == 0. */
int st_conv3x3_fwd_bf16_g8_tap_pool(const void* x_g8, const void* w_packed, const float* bias, float* y,
                                    void* y_pooled_g8, int n, int cin, int hin, int win, int cout, int pad_mode,
                                    int relu, st_stream_t stream);
/* ... and its backward half: the G8 gradient w.r.t. the pooled activation [n][c/8][h/2][w/2][8] routed to the
 * position the code names where the code's bit 2 is set, zeros elsewhere -> G8 [n][c/8][h][w][8], the (already ReLU-
 * gated) operand of st_conv3x3_dgrad_bf16_g8.  Same values as st_grad_join_g8(chain_pooled, x_gate) on the
 * full-resolution activation. */
int st_grad_pool_codes_g8(const void* g_pooled_g8, const void* codes, void* out_g8, int n, int c, int h, int w,
                          st_stream_t stream);

/* backward: x is the pool INPUT [nc][h][w], dy [nc][ho][wo]; dx [nc][h][w] fully overwritten.
 * Ties go to the first maximum in scan order, as torch.max_pool2d does. */
int st_maxpool2x2_bwd(const float* x, const float* dy, float* dx, int nc, int h, int w,
                      int ceil_mode, st_stream_t stream);

/* ---- AdaIN (transformer.py:15-24, 46-59) + alpha blend (trainer.py:434, :702) ---------
 * out = alpha * ((content - mu_c)/sigma_c * sigma_s + mu_s) + (1 - alpha) * content
 * sigma = sqrt(var_unbiased + eps), per (n, c) plane; content/style may differ in H*W.
 * stats (nullable): [n*c][4] = {mu_c, sigma_c, mu_s, sigma_s} saved for backward. */
int st_adain_fwd(const float* content, const float* style, float* out,
                 int n, int c, int hw_content, int hw_style,
                 float eps_content, float eps_style, float alpha,
                 float* stats, st_stream_t stream);

/* Same, with the two eps values read from DEVICE memory (eps_dev[0] = eps_content, eps_dev[1] =
 * eps_style): the trainable eps of transformer.py:11-12, 27-44 never has to be read back by the host, so
 * a training / inference step has no host-device synchronisation in it (the reference's adjust_sigmoid
 * reads the 1-element tensors on the host).  The caller guarantees 0 < eps < 1 (clamped by construction). */
int st_adain_fwd_dev_eps(const float* content, const float* style, float* out,
                         int n, int c, int hw_content, int hw_style,
                         const float* eps_dev, float alpha, float* stats, st_stream_t stream);

/* AdaIN backward.  gout: [n][c][hw_content] gradient of `out`; stats: as saved by st_adain_fwd.
 * dcontent / dstyle: nullable (the training loop needs neither: the encoder is frozen).
 * deps[2] = {d/d eps_content, d/d eps_style} (the trainable eps of transformer.py:11-12, before
 * the sqrt(|p|) of adjust_sigmoid).  workspace: n*c*2 floats. */
int st_adain_bwd(const float* gout, const float* content, const float* style, const float* stats,
                 float alpha, float* dcontent, float* dstyle, float* deps, float* workspace,
                 int n, int c, int hw_content, int hw_style, st_stream_t stream);

/* per-plane mean and sqrt(unbiased var + eps)  (AdaIN.calc_mean_std transformer.py:15-24,
 * AdaINStyleLoss.calc_mean_std losses.py:20-28).  mean, std: [planes] */
int st_meanstd_fwd(const float* x, float* mean, float* std, int planes, int hw, float eps,
                   st_stream_t stream);

/* ---- loss reductions (fixed-order two-stage sums; `workspace` >= st_reduce_workspace_bytes()).
 * Backward entry points take the upstream gradient as a DEVICE scalar `gout` (NULL = 1) times
 * the host `scale`, and overwrite (accumulate = 0) or add into (accumulate = 1) dx. ------------ */
size_t st_reduce_workspace_bytes(void);

/* nn.MSELoss: AdaINContentLoss.forward (losses.py:93-101).  out[0] = mean((a-b)^2);
 * da (+)= gout*scale * 2 (a-b) / n. */
int st_mse_fwd(const float* a, const float* b, long long n, float* workspace, float* out,
               st_stream_t stream);
int st_mse_bwd(const float* a, const float* b, const float* gout, float scale, float* da,
               long long n, int accumulate, st_stream_t stream);

/* TVLoss.forward (losses.py:104-116) on x [n][c][h][w]. */
int st_tv_fwd(const float* x, int n, int c, int h, int w, float* workspace, float* out,
              st_stream_t stream);
int st_tv_bwd(const float* x, const float* gout, float scale, float* dx, int n, int c, int h, int w,
              int accumulate, st_stream_t stream);

/* AdaINStyleLoss.calc_style_loss (losses.py:30-36) from per-plane stats (st_meanstd_fwd, eps 1e-5):
 * out[0] = mean((mean_a-mean_b)^2) + mean((std_a-std_b)^2); bwd w.r.t. the feature map x that
 * produced (mean_a, std_a). */
int st_meanstd_loss_fwd(const float* mean_a, const float* std_a, const float* mean_b,
                        const float* std_b, int planes, float* workspace, float* out,
                        st_stream_t stream);
int st_meanstd_loss_bwd(const float* x, const float* mean_a, const float* std_a, const float* mean_b,
                        const float* std_b, const float* gout, float scale, float* dx, int planes,
                        int hw, int accumulate, st_stream_t stream);
/* The per-plane scalars of that backward, coeffs [planes][4] = (k0, k1, mean_a, 0) with dx = k0 + k1 * (x - mean_a):
 * for st_grad_join_g8_affine. */
int st_meanstd_loss_bwd_coeffs(const float* mean_a, const float* std_a, const float* mean_b, const float* std_b,
                               const float* gout, float scale, float* coeffs, int planes, int hw, st_stream_t stream);

/* ---- RGB-uv histogram (RGBuvHistBlock.forward, losses.py:169-322; 'interpolation' resizing,
 * all three pairings, h <= 64) and the Hellinger distance of HistLoss (losses.py:338-347) -------
 *   x:      [n][c>=3][h][w] (first three channels are used, losses.py:192-193)
 *   hist:   [n][3][64][64] normalised (each image sums to 1); totals: [n] un-normalised sums
 *   bins_host: HOST array of nbins (1..64) float64 bin centres, np.linspace(lo, hi, nbins); with nbins < 64 the
 *           histogram occupies hist[n][k][:nbins][:nbins] and the rest of each 64 x 64 plane is zero
 *           ('sampling' resizing and green_only are host-side selections around this call: losses.RGBuvHistBlock)
 *   method: ST_HIST_*; sigma as in losses.py:124; insz: resize threshold/size (HistLoss: 128)
 * In ST_HIST_THRESHOLDING with intensity_scale = 0 the un-normalised bins are exact integer
 * counts (no atomics, exact fp32 sums).  workspace: st_rgbuv_hist_workspace_bytes(). */
size_t st_rgbuv_hist_workspace_bytes(int n, int h, int w, int insz);
int st_rgbuv_hist_fwd(const float* x, float* hist, float* totals, float* workspace, int n, int c,
                      int h, int w, int insz, int method, float sigma, const double* bins_host,
                      int nbins, int intensity_scale, st_stream_t stream);
/* The same forward for callers that CANNOT ALLOCATE (code under HIP-graph capture, tight serving loops, harnesses
 * that forbid hipMalloc): no workspace argument -- the pixel records and the split slabs live in a file-scope
 * __device__ array inside the library, ST_HIST_STATIC_SCRATCH_BYTES per GPU the library runs on, held until the
 * process ends.  This entry point has NO counterpart in the reference; same kernels and bit-identical results as
 * st_rgbuv_hist_fwd.  It calls no hipMalloc / hipFree or any allocating routine; the first call on a device
 * resolves the array's device address (hipGetSymbolAddress, that device must be current), so make that first call
 * outside a stream capture.  Shapes whose workspace exceeds the buffer are rejected with ST_ERR_BAD_ARG (HistLoss
 * shapes: any batch of up to 32 images per call; st_rgbuv_hist_static_scratch_bytes() reports the limit).
 * Every call on one GPU shares that GPU's buffer: calls that overlap on one GPU (two streams, two threads)
 * race -- serialise them.  Nothing in the buffer has to start at zero: each call writes everything it reads. */
#define ST_HIST_STATIC_SCRATCH_BYTES (32u << 20)
size_t st_rgbuv_hist_static_scratch_bytes(void);
int st_rgbuv_hist_fwd_static(const float* x, float* hist, float* totals, int n, int c, int h, int w, int insz,
                             int method, float sigma, const double* bins_host, int nbins,
                             int intensity_scale, st_stream_t stream);
/* dx (+)= gout*scale * d<ghist, hist>/dx, through normalisation, kernel weights, log-chroma,
 * intensity, bilinear resize and clamp.  hist/totals: the forward outputs. */
int st_rgbuv_hist_bwd(const float* x, const float* hist, const float* totals, const float* ghist,
                      const float* gout, float scale, float* dx, float* workspace, int n, int c,
                      int h, int w, int insz, int method, float sigma, const double* bins_host,
                      int nbins, int intensity_scale, int accumulate, st_stream_t stream);

/* out[0] = sum_k (sqrt(h_target[k]) - sqrt(h_input[k]))^2 over the whole batch (the quantity under
 * the outer sqrt of losses.py:342-344; data-parallel training all-reduces exactly this scalar).
 * bwd: dh_input = gq*scale * d out / d h_input.  workspace >= st_reduce_workspace_bytes(). */
int st_hellinger_sq_fwd(const float* h_target, const float* h_input, long long n, float* workspace,
                        float* out, st_stream_t stream);
int st_hellinger_sq_bwd(const float* h_target, const float* h_input, const float* gq, float scale,
                        float* dh_input, long long n, st_stream_t stream);

/* ---- un-normalised covariance of VincentStyleLoss.calc_2_moments (losses.py:49-54) -----------
 * x: [b][c][n] (n = H*W), mean: [b][c] (st_meanstd_fwd), cov: [b][c][c] = (x-mean)(x-mean)^T.
 * MFMA GEMM over pixels, split-K slabs in `workspace` (st_cov_workspace_bytes), fixed-order sum.
 * bwd: dx = (gcov + gcov^T)(x - mean) + gmean / n   (gmean [b][c] nullable).
 * The eigendecompositions of l2wass_dist (losses.py:67-81) stay on torch.linalg.eigh. */
size_t st_cov_workspace_bytes(int b, int c, int n);
int st_cov_fwd(const float* x, const float* mean, float* cov, float* workspace, int b, int c, int n,
               st_stream_t stream);
int st_cov_bwd(const float* x, const float* mean, const float* gcov, const float* gmean, float* dx,
               int b, int c, int n, st_stream_t stream);

/* ---- batched symmetric eigendecomposition of small matrices: torch.linalg.eigh / eigvalsh inside
 * VincentStyleLoss.l2wass_dist (losses.py:69, 73, 78) for the style layers with <= 128 channels ---------------
 * a: [batch][c][c] symmetric (both triangles are read and averaged); w: [batch][c] eigenvalues in ascending
 * order; v: [batch][c][c] row-major, COLUMN j is the eigenvector of w[j] (a = v diag(w) v^T); sweeps: nullable,
 * [batch] number of Jacobi sweeps used; max_sweeps + 1 means "did not converge" (the analogue of LAPACK's info).
 * One workgroup per matrix, matrix and eigenvectors resident in LDS, parallel-ordered cyclic Jacobi; c even,
 * 2 <= c <= st_syevj_small_max_dim() (128).  Larger matrices stay on rocSOLVER's syevd (style_transfer_amd/eigh.py). */
int st_syevj_small_max_dim(void);
int st_syevj_small_batched(const float* a, float* w, float* v, int* sweeps, int batch, int c, int max_sweeps,
                           st_stream_t stream);

/* ---- GPU-side training augmentations: the per-image tensor transforms composed after ToTensor
 * in src/train.py:116-135, applied to a batch x: [n][3][h][w] in HBM with per-image parameters in small DEVICE
 * arrays (drawn on the host).  y is a second buffer of the same shape unless stated; every call is one or a few
 * single-pass launches.  Values follow style_transfer_amd/transforms.py, the torchvision-free restatement of the
 * same transforms (torchvision 0.15.1 semantics; the two custom transforms: src/utils/custom_transform.py:9-46). */
/* RGBToGrayscaleStacked(enable=True), custom_transform.py:32-46 (y may alias x). */
int st_aug_gray_stack(const float* x, float* y, int n, int hw, st_stream_t stream);
/* transforms.RandomRotation: nearest, same canvas, zero fill; angle_rad_dev[n] (0 = identity). */
int st_aug_rotate(const float* x, float* y, const float* angle_rad_dev, int n, int c, int h, int w,
                  st_stream_t stream);
/* AddGaussianNoise, custom_transform.py:9-29: images with sigma_dev[i] >= 0 become
 * floor(interp(interp(x, (min, max), (0, 1)) + noise, (min', max'), (0, 255))) / 255, the others are copied.
 * noise: [n][elems] samples of N(mean, sigma_i) (the caller's RNG); minmax_ws: 2 n floats; y may alias x. */
int st_aug_gaussian_noise(const float* x, const float* noise, const float* sigma_dev, float* y,
                          float* minmax_ws, int n, int elems_per_image, st_stream_t stream);
/* transforms.RandomAdjustSharpness: factor_dev[i] < 0 = image not selected. */
int st_aug_sharpness(const float* x, float* y, const float* factor_dev, int n, int c, int h, int w,
                     st_stream_t stream);
/* One of ColorJitter's four adjustments per call, chosen per image: code_dev[i] = 0 brightness, 1 contrast,
 * 2 saturation, 3 hue, other = copy; param_dev[i] = the factor (hue: the shift).  Four calls with the four columns of
 * a per-image permutation reproduce ColorJitter's random order.  gray_mean_ws: n floats; y may alias x. */
int st_aug_color_jitter_slot(const float* x, float* y, const int* code_dev, const float* param_dev,
                             float* gray_mean_ws, int n, int hw, st_stream_t stream);
/* transforms.RandomHorizontalFlip: flip_dev[i] != 0 = flip. */
int st_aug_hflip(const float* x, float* y, const int* flip_dev, int n, int c, int h, int w, st_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ST_HIP_H */
