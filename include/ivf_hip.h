/* libivf_hip.so -- C-ABI of the MI355X (gfx950) video-saliency hot path.
 *
 * The reference (interpreting-video-features) has no FFI/plugin boundary: its hot
 * path is plain PyTorch called from Python.  This header is therefore the
 * build-defined boundary beneath the reference's Python call surface (SURVEY.md
 * section 8b); each entry point names the reference code it replaces.  File:line
 * citations are relative to /root/reference/video_features_pytorch/ ("smth" =
 * FindMasksComparison_I3D_smth.py).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 unless its name ends in _host or
 *    its type says otherwise; the caller owns all memory, the library never
 *    allocates or frees device memory; scratch is passed in, sized by the
 *    matching *_workspace_bytes() query;
 *  - `stream` is a hipStream_t (0 = the null stream); every call only enqueues
 *    work and never synchronises, so a call sequence can be captured in a hipGraph;
 *  - return value 0 = ok, negative = error (IVF_ERR_*), message via
 *    ivf_last_error() (thread-local);
 *  - activations inside the library are channels-last [B, T, H, W, ld] fp32 with a
 *    channel window (coff, C) inside the row of ld floats; reference tensors are
 *    NCTHW [B, C, T, H, W] and are converted at the edges (freeze/reverse kernels
 *    write channels-last directly).
 */
#ifndef IVF_HIP_H
#define IVF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IVF_OK 0
#define IVF_ERR_BAD_ARG (-1)
#define IVF_ERR_HIP (-2)
#define IVF_ERR_UNSUPPORTED (-3)

typedef void* ivf_stream_t; /* hipStream_t */

const char* ivf_last_error(void);
int ivf_version(void);

/* ------------------------------------------------------------------ mask.py */

/* mask.perturb_sequence(..., 'freeze') forward, mask.py:11-22.
 * x [B,C,T,HW] NCTHW; mask [T] (mask_per_clip=0) or [B,T] (=1), already in [0,1];
 * p: out_cpad==0 -> NCTHW, else channels-last [B,T,HW,out_cpad] (pad channels zero). */
int ivf_freeze_fwd(const float* x, const float* mask, float* p, int B, int C, int T, int HW,
                   int mask_per_clip, int out_cpad, ivf_stream_t stream);

/* Backward of the same (autograd of mask.py:11-22 under smth:213).
 * g: upstream gradient, NCTHW (g_cpad==0) or channels-last with row g_cpad.
 * dmask [B,T] per-clip gradient (sum over B yourself for a shared mask; entry 0 is 0);
 * dx NCTHW or NULL.  Deterministic two-stage reduction (no float atomics). */
size_t ivf_freeze_bwd_workspace_bytes(int B, int T);
int ivf_freeze_bwd(const float* x, const float* mask, const float* g, float* dmask, float* dx, int B,
                   int C, int T, int HW, int mask_per_clip, int g_cpad, void* workspace,
                   ivf_stream_t stream);

/* find_submasks_from_mask, mask.py:60-85, plus the pairing rule of the 'reverse'
 * perturbation, mask.py:40-56.  run[t] = index of the run containing frame t or -1;
 * partner[t] = frame swapped with t (t itself if copied); weight[t] = mask value of
 * the pair's first-half member (used for BOTH members, mask.py:50-56). */
int ivf_submask_pairs(const float* mask, int T, float thresh, int* run, int* partner, float* weight,
                      ivf_stream_t stream);

/* mask.perturb_sequence(..., 'reverse') given the pairing above. */
int ivf_reverse_fwd(const float* x, const int* partner, const float* weight, float* p, int B, int C,
                    int T, int HW, int out_cpad, ivf_stream_t stream);

/* The same for b clips with per-clip masks [B,T] (partner/weight rows [B,T]); forward writes
 * NCTHW (out_cpad 0) or 16-byte channels-last pixels (out_cpad 4). */
int ivf_submask_pairs_batched(const float* mask, int B, int T, float thresh, int* partner, float* weight,
                              ivf_stream_t stream);
int ivf_reverse_fwd_batched(const float* x, const int* partner, const float* weight, float* p, int B, int C,
                            int T, int HW, int out_cpad, ivf_stream_t stream);
/* Autograd of mask.py:49-56 w.r.t. the mask (the reference optimises through 'reverse' when
 * temporalMaskType == 'reverse', smth:121,202): for a pair (a, b'), a in the first half of its run,
 * dmask[a] = sum (X[b'] - X[a]) * (G[a] - G[b']); all other entries 0.  g NCTHW (g_cpad 0) or
 * channels-last; workspace: ivf_freeze_bwd_workspace_bytes(B, T). */
int ivf_reverse_bwd(const float* x, const int* partner, const float* g, float* dmask, int B, int C, int T,
                    int HW, int g_cpad, void* workspace, ivf_stream_t stream);

/* mask.calc_tv_norm(mask, p, q), mask.py:88-100: val[b] and (optional) grad[b,:]. */
int ivf_tv_norm(const float* mask, int B, int T, float p, float q, float* val, float* grad,
                ivf_stream_t stream);

/* Regulariser of the search loop, smth:198-200: sig = sigmoid(raw);
 * terms[b] = {lam1*sum|sig|, lam2*TV33(sig)}; dreg_dsig = d(l1+tv)/dsig. */
int ivf_mask_reg(const float* raw_mask, int B, int T, float lam1, float lam2, float* sig, float* terms,
                 float* dreg_dsig, ivf_stream_t stream);

/* torch.optim.Adam step on one tensor (smth:191,214); step counts from 1. */
int ivf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int n, int step,
                  float lr, float beta1, float beta2, float eps, ivf_stream_t stream);

/* Loop tail, smth:207-214: chain through the sigmoid, Adam step on raw_mask [B,T],
 * traj_row[b] = (loss, l1, tv, score). */
int ivf_search_step(float* raw_mask, const float* sig, const float* dscore_dsig, const float* dreg_dsig,
                    const float* terms, const float* score, float* exp_avg, float* exp_avg_sq,
                    float* traj_row, int B, int T, int step, float lr, float beta1, float beta2,
                    float eps, ivf_stream_t stream);

int ivf_sigmoid(const float* x, float* y, int n, ivf_stream_t stream);

/* Integer frame-importance ranking of finished masks (the drivers rank frames by mask value,
 * FindMasksComparison_I3D_smth.py:216-230): order[b][r] = frame with the r-th largest value of mask[b][:], ties by
 * frame index, NaN last -- torch.argsort(-mask, stable=True).  mask [B,T] fp32, order [B,T] int32. */
int ivf_rank_frames(const float* mask, int B, int T, int* order, ivf_stream_t stream);

/* init_mask(mode='central') selection for B clips (mask.py:134-154): orig[b] / full[b] = target score of the clip /
 * of its fully frozen version, central[b][i] = score under candidate i+1 (ones with i+1 zeros at each end), n
 * candidates.  ratio[b][i] = (orig - central) / (orig - full) (optional output); chosen_i[b] (optional, 1-based) = first
 * candidate whose ratio < threshold, else n (NaN compares false, as in the reference); raw_mask[b][t] = -5 on the
 * chosen candidate's zeros, +5 on its ones. */
int ivf_init_central_select(const float* orig, const float* full, const float* central, int B, int n, int T,
                            float threshold, float* raw_mask, int* chosen_i, float* ratio, ivf_stream_t stream);

/* Clip ingest: the arithmetic of ImLoader.__getitem__ /
 * KTHImLoader.__getitem__ after the JPEG decode (data_loader_jpg.py:29-37,
 * data_loader_kth.py:25-44): uint8 frames [B][T][H][W][C] -> float32 (exact), permuted to
 * [B][C][T][H][W] (layout IVF_INGEST_NCTHW, the tensor the reference hands to the model) or
 * to channels-last rows of cpad >= C floats with zero pad lanes (IVF_INGEST_CL, the plan's
 * input layout).  The host uploads a quarter of the bytes of the fp32 clip. */
#define IVF_INGEST_NCTHW 0
#define IVF_INGEST_CL 1
int ivf_clip_ingest_u8(const unsigned char* frames, float* out, int B, int T, int H, int W, int C, int layout,
                       int cpad, ivf_stream_t stream);

/* ------------------------------------------------------------------ Unit3D */

/* Arithmetic of the implicit-GEMM convolution:
 *  IVF_MATH_FP32    v_mfma_f32_32x32x2_f32: exact fp32 FMA chains (157 TFLOP/s peak);
 *  IVF_MATH_BF16X3  every operand split x = hi + lo (two bf16), three
 *                   v_mfma_f32_32x32x16_bf16 per k-step (lo*hi + hi*lo + hi*hi), fp32
 *                   accumulate: ~2^-17 relative per product, 5.3x the fp32-MFMA rate;
 *  IVF_MATH_BF16X6  every operand split x = hi + mid + lo (three bf16 = all 24 bits of the fp32
 *                   significand), six MFMAs per k-step (every term down to 2^-16 of the product), fp32
 *                   accumulate: an fp32 product to ~2^-23 -- the reference's fp32 arithmetic on the bf16
 *                   matrix cores, 2.65x the fp32-MFMA rate (ceiling 417 TFLOP/s);
 *  IVF_MATH_BF16ACT activations AND gradients are stored as bf16 in HBM (round-to-nearest-even in the
 *                   epilogues), weights stay split hi/lo, two MFMAs per k-step (a*lo + a*hi), fp32
 *                   accumulate (BASELINE configs[4]: "bf16 activations").  In this mode every in / out /
 *                   in2 / out2 / relu_mask pointer of ivf_conv3d addresses bf16 elements (same element
 *                   offsets), with two exceptions at the ends of the network: a 4-channel-pixel strided
 *                   convolution (IVF_CONV_PIX4, the stem) READS fp32 pixels, and a depth-to-space
 *                   backward (d2s, the stem's input gradient) WRITES fp32. */
#define IVF_MATH_FP32 0
#define IVF_MATH_BF16X3 1
#define IVF_MATH_BF16X6 2
#define IVF_MATH_BF16ACT 3

/* One convolution as implicit GEMM on the matrix cores.  Forward of Unit3D
 * (models/I3D_doubled.py:83-118: asymmetric zero pad + Conv3d + BN(eval) + ReLU) and,
 * with the packed backward weights, its backward-data.  Channels-last in/out. */
typedef struct {
  int B, Ti, Hi, Wi;           /* input positions */
  int Cin, in_ld, in_coff;     /* channel window read (Cin % 4 == 0) */
  int To, Ho, Wo;              /* output positions (block grid when d2s) */
  int Cout, out_ld, out_coff;  /* rows of the packed weight / channel window written */
  int kT, kH, kW, sT, sH, sW;  /* kernel, stride */
  int pT, pH, pW;              /* FRONT zero padding; back padding is implied by To/Ho/Wo */
  int relu;                    /* max(.,0) in the epilogue */
  int accumulate;              /* out += result (before relu / mask) */
  int mask_ld, mask_coff;      /* geometry of relu_mask (same positions as out) */
  int d2s;                     /* depth-to-space output: stride-2 backward-data */
  int dT, dH, dW, dC;          /* d2s: real output dims and channels written */
  int bsT, bsH, bsW;           /* d2s: block strides (forward strides, 1 or 2) */
  int math;                    /* IVF_MATH_*; must match the weight pack */
  int variant;                 /* IVF_CONV_AUTO, or a kernel variant id from ivf_conv3d_variants() */
  /* optional second input of a 1x1x1 conv: GEMM-K channels [K0, Cin) are read from in2 (same
   * positions; row length in2_ld, channel offset in2_coff), channels [0, K0) from `in` */
  int K0, in2_ld, in2_coff;
  const float* in2;
  /* optional second OUTPUT window (forward epilogues only: no accumulate, no relu_mask, no d2s): columns
   * [N0, Cout) are written to out2 (row length out2_ld, channel offset out2_coff, column n - N0) instead of
   * `out` -- one GEMM for several units that read the same input and write different buffers (an Inception
   * module's b0 | b1a | b2a).  Served by the implicit-GEMM tiles only. */
  int N0, out2_ld, out2_coff;
  float* out2;
  /* 1-bit ReLU gates (1 byte per 8 channels, rows of gate_*_ld BYTES; channel offsets are multiples of 8 and so is
   * Cout / N0).  A forward epilogue can record (value > 0) of what it stores: gate_out for the `out` window,
   * gate_out2 for the `out2` window.  A backward epilogue can take its gate from such a record (gate_in) instead
   * of re-reading the fp32 activation through `relu_mask` -- 1/32 of the bytes.  All optional (null = off). */
  unsigned char* gate_out;
  unsigned char* gate_out2;
  const unsigned char* gate_in;
  int gate_out_ld, gate_out_coff, gate_out2_ld, gate_in_ld, gate_in_coff;
} ivf_conv3d_desc;

/* Kernel variants: tile shapes of the plain implicit GEMM (IVF_CONV_IGEMM_BASE + 0..2) and of
 * the LDS-halo kernel (IVF_CONV_HALO_BASE + i; split-bf16, stride 1, 2 <= k <= 4 only).  All
 * variants compute the same sums; they differ in speed per layer shape and (in the last
 * bits) in summation order, so a plan fixes one variant per layer. */
#define IVF_CONV_AUTO 0
#define IVF_CONV_IGEMM_BASE 1
#define IVF_CONV_HALO_BASE 16
#define IVF_CONV_PIX4 15 /* 4-channel-pixel strided kernel (the stem): split-bf16, Cin = in_ld = 4, stride (1|2, 2, 2), k <= 7 */
int ivf_conv3d_variants(const ivf_conv3d_desc* d, int* ids, int max_ids);

/* out = epilogue(conv(in, w_packed)): v = acc*scale[n] + shift[n] (NULL = 1 / 0);
 * relu_mask != NULL zeroes v where mask <= 0 (the ReLU below, for backward-data). */
int ivf_conv3d(const ivf_conv3d_desc* d, const float* in, const float* w_packed, const float* scale,
               const float* shift, const float* relu_mask, float* out, ivf_stream_t stream);

/* BatchNorm3d(eval) fold, I3D_doubled.py:75 (eps 1e-3). */
int ivf_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                float* scale, float* shift, int C, ivf_stream_t stream);

/* Reference weight [Cout][Cin][kT][kH][kW] -> forward pack [Cout][taps*CinPad]
 * (fp32, or bf16 planes with rows padded to 8: hi/lo for IVF_MATH_BF16X3 and IVF_MATH_BF16ACT, hi/mid/lo for
 * IVF_MATH_BF16X6);
 * w_packed holds ivf_conv3d_pack_fwd_elems() floats. */
size_t ivf_conv3d_pack_fwd_elems(int Cout, int CinPad, int kT, int kH, int kW, int math);
int ivf_conv3d_pack_fwd(const float* w_ref, float* w_packed, int Cout, int Cin, int CinPad, int kT,
                        int kH, int kW, int math, ivf_stream_t stream);
/* Same, as rows [row0, row0+Cout) of a packed matrix with rows_total rows: several units that
 * read the same input packed side by side so they run as one convolution. */
int ivf_conv3d_pack_fwd_rows(const float* w_ref, float* w_packed, int Cout, int Cin, int CinPad, int kT,
                             int kH, int kW, int row0, int rows_total, int math, ivf_stream_t stream);

/* Geometry of the backward-data convolution produced by ivf_conv3d_pack_bwd. */
typedef struct {
  int d2s;          /* 0: plain flipped conv (all strides 1); 1: depth-to-space form */
  int kT, kH, kW;   /* kernel of the backward conv over dY */
  int pT, pH, pW;   /* its front padding */
  int rows;         /* rows of the packed weight (= Cout of the backward conv) */
} ivf_conv3d_bwd_geom;

/* Backward-data pack, BN scale folded in: [rows][taps_b*Cout]; needs
 * rows*taps_b*Cout floats where taps_b = geom.kT*kH*kW (call with w_packed sized
 * by ivf_conv3d_pack_bwd_elems). */
size_t ivf_conv3d_pack_bwd_elems(int Cout, int CinPad, int kT, int kH, int kW, int sT, int sH, int sW,
                                 int pT, int pH, int pW, int math);
int ivf_conv3d_pack_bwd(const float* w_ref, const float* scale, float* w_packed, int Cout, int Cin,
                        int CinPad, int kT, int kH, int kW, int sT, int sH, int sW, int pT, int pH,
                        int pW, int math, ivf_conv3d_bwd_geom* geom, ivf_stream_t stream);

/* The 1x1x1 units b0 / b1a / b2a of an Inception module read the same input, so their
 * backward-data is ONE GEMM over the concatenated gradients [dY_b0 | dT_b1a | dT_b2a]:
 * pack unit by unit, in any order, into a [CinPad][Ktotal] matrix (column offset koff; the
 * unit ending at Ktotal also zeroes the row padding). */
size_t ivf_conv3d_pack_bwd_fused1x1_elems(int Ktotal, int CinPad, int math);
int ivf_conv3d_pack_bwd_fused1x1(const float* w_ref, const float* scale, float* w_packed, int Cout, int Cin,
                                 int CinPad, int koff, int Ktotal, int math, ivf_stream_t stream);

/* ------------------------------------------------------------------ pooling / head */

typedef struct {
  int B, Ti, Hi, Wi, C, in_ld, in_coff;
  int To, Ho, Wo, out_ld, out_coff;
  int kT, kH, kW, sT, sH, sW, pT, pH, pW; /* front ZERO padding (I3D_doubled.py:36-39) */
  /* forward only: record arg-max 255 ("no route") for windows whose maximum is not > 0.
   * When x is a ReLU output and the gradient is wanted below that ReLU, the backward then
   * needs no relu_mask: a cell with x <= 0 can only win a window whose maximum is <= 0. */
  int gate_nonpos;
  /* storage of x / y / dy / dx / relu_mask: 0 = float, 1 = bf16 (IVF_MATH_BF16ACT plans; same element offsets).  The
   * forward only selects, so it is exact in either storage; the backward sums in fp32 and rounds once on store. */
  int act_bf16;
} ivf_pool3d_desc;

/* MaxPool3dSamePadding.forward, I3D_doubled.py:15-40; argmax [positions_out][C] uint8
 * (flat tap of the winner, first strict maximum, pad cells count as 0). */
int ivf_maxpool3d_fwd(const ivf_pool3d_desc* d, const void* x, void* y, unsigned char* argmax,
                      ivf_stream_t stream);
int ivf_maxpool3d_bwd(const ivf_pool3d_desc* d, const void* dy, const unsigned char* argmax, void* dx,
                      const void* relu_mask, int accumulate, ivf_stream_t stream);

/* I3D head, I3D_doubled.py:360-380, for a pooling window covering the whole feature
 * map: feat [B,npos,C] -> pooled [B,C] (optional) -> logits [B,K] -> probs [B,K]
 * (softmax over K if softmax != 0, else a copy).  w is the reference
 * logits.conv3d.weight viewed as [K][C]. */
int ivf_head_fwd(const float* feat, const float* w, const float* bias, float* pooled, float* logits,
                 float* probs, int B, int npos, int C, int K, int softmax, ivf_stream_t stream);

/* Backward of the head.  Upstream gradient: dout [B,K] if non-NULL, else one-hot at
 * target[b].  score[b] = probs[b,target[b]] (optional, needs target).
 * dpooled [B,C] optional; dfeat [B,npos,C] optional, gated by feat > 0 when gate_relu. */
int ivf_head_bwd(const float* feat, const float* w, const float* probs, const int* target,
                 const float* dout, float* score, float* dpooled, float* dfeat, int B, int npos, int C,
                 int K, int softmax, int gate_relu, ivf_stream_t stream);
/* The same two with the feature map (and dfeat) stored as bf16 (IVF_MATH_BF16ACT plans); pooled / logits /
 * probs / score / dpooled stay fp32. */
int ivf_head_fwd_bf16(const void* feat, const float* w, const float* bias, float* pooled, float* logits,
                      float* probs, int B, int npos, int C, int K, int softmax, ivf_stream_t stream);
int ivf_head_bwd_bf16(const void* feat, const float* w, const float* probs, const int* target,
                      const float* dout, float* score, float* dpooled, void* dfeat, int B, int npos, int C,
                      int K, int softmax, int gate_relu, ivf_stream_t stream);

/* ------------------------------------------------------------------ Grad-CAM */

/* grad_cam_videos.py:98-110: weights[b,k] = mean_pos grad; cam[b,pos] = relu(sum_k w*feat). */
int ivf_gradcam_reduce(const float* feat, const float* grad, float* weights, float* cam, int B, int npos,
                       int C, ivf_stream_t stream);
/* feat and grad stored as bf16 (IVF_MATH_BF16ACT plans); weights and cam fp32. */
int ivf_gradcam_reduce_bf16(const void* feat, const void* grad, float* weights, float* cam, int B, int npos,
                            int C, ivf_stream_t stream);

/* grad_cam_videos.py:113-138: per temporal slice bilinear resize (sh,sw)->(H,W)
 * (OpenCV INTER_LINEAR rule), repeat `step` frames, min/max normalise per slice block
 * (per_frame) or per clip.  cam [B,nslice,sh,sw] -> out [B,nslice*step,H,W];
 * minmax_ws: B*nslice*2 floats. */
int ivf_cam_resize_normalise(const float* cam, float* out, float* minmax_ws, int B, int nslice, int sh,
                             int sw, int H, int W, int step, int per_frame, ivf_stream_t stream);

/* ------------------------------------------------------------------ whole I3D */

typedef struct {
  int B;                 /* maximum clips per call */
  int C, T, H, W;        /* clip geometry, reference NCTHW */
  int num_classes;
  int stem_stride_t;     /* temporal stride of Conv3d_1a_7x7 (2, or last_stride) */
  int pool4a_stride_t;   /* MaxPool3d_4a_3x3 */
  int pool5a_stride_t;   /* MaxPool3d_5a_2x2 */
  int head_kt, head_kh, head_kw; /* AvgPool3d window: (2,7,7) / (finalTimeLength,4,5) */
  int softmax;           /* Model(softMax=...) */
  int math;              /* IVF_MATH_* for every Unit3D convolution (IVF_MATH_BF16ACT also selects bf16 storage of
                            every activation / gradient buffer of the plan except the clip and its gradient) */
} ivf_i3d_config;

typedef struct ivf_i3d ivf_i3d_t;

/* Host-side plan (shapes, buffer offsets, launch list).  No device work. */
int ivf_i3d_create(const ivf_i3d_config* cfg, ivf_i3d_t** out);
/* Run the HBM-bound branch of every Inception module (the 3x3x3 pool and b3b, both ways) on a side
 * stream beside the 3x3x3 convs of the other branches (fork at module entry, join before the next consumer, with
 * events inside every forward / backward call; results are bit-identical).  On by default (IVF_OVERLAP=0 in the
 * environment of ivf_i3d_create turns it off): +1.2 % measured, see DESIGN.md. */
int ivf_i3d_set_overlap(ivf_i3d_t* net, int on);
void ivf_i3d_destroy(ivf_i3d_t* net);
size_t ivf_i3d_weights_bytes(const ivf_i3d_t* net);
size_t ivf_i3d_workspace_bytes(const ivf_i3d_t* net);
/* Give the plan its two caller-owned device arenas (256-byte aligned). */
int ivf_i3d_bind(ivf_i3d_t* net, void* weights_arena, void* workspace);

/* Convolution units in reference registration order (I3D_doubled.py:229-334);
 * name is the state_dict prefix, e.g. "Mixed_3b.b1a" or "logits". */
int ivf_i3d_num_convs(const ivf_i3d_t* net);
int ivf_i3d_conv_info(const ivf_i3d_t* net, int i, char* name64, int* cout, int* cin, int* kT, int* kH,
                      int* kW, int* has_bn);
/* Pack one unit from reference-layout device tensors: w [Cout][Cin][kT][kH][kW];
 * BN tensors (NULL for the logits unit, which takes bias instead). */
int ivf_i3d_load_conv(ivf_i3d_t* net, int i, const float* w, const float* bn_gamma, const float* bn_beta,
                      const float* bn_mean, const float* bn_var, const float* bias, float bn_eps,
                      ivf_stream_t stream);

/* Model.forward, I3D_doubled.py:351-380, on b <= B clips.  x NCTHW.
 * logits/probs [b,K] outputs (either may be NULL); activations stay in the workspace. */
int ivf_i3d_forward(ivf_i3d_t* net, const float* x, int b, float* logits, float* probs,
                    ivf_stream_t stream);
/* Forward of an already perturbed channels-last clip held in the plan's input buffer. */
int ivf_i3d_forward_staged(ivf_i3d_t* net, int b, float* logits, float* probs, ivf_stream_t stream);
/* Device pointer of the plan's input buffer [B,T,H*W,4] (channels-last, C padded to 4). */
float* ivf_i3d_input_buffer(ivf_i3d_t* net);
/* Backward-data of the last forward: upstream dout [b,K] or one-hot target[b];
 * writes score[b] (optional), dx NCTHW (optional); the channels-last gradient of
 * the input stays in the workspace (ivf_i3d_input_grad_buffer). */
int ivf_i3d_backward(ivf_i3d_t* net, int b, const int* target, const float* dout, float* score,
                     float* dx, ivf_stream_t stream);
float* ivf_i3d_input_grad_buffer(ivf_i3d_t* net);

/* Named endpoint activations of the last forward (tests, Grad-CAM):
 * channels-last [b,T,H,W,ld]; returns IVF_ERR_BAD_ARG for an unknown name. */
int ivf_i3d_endpoint(const ivf_i3d_t* net, const char* name, float** ptr, int* T, int* H, int* W, int* C,
                     int* ld);
/* Bytes per stored element of those endpoints (and of every other activation / gradient buffer except the clip and
 * its gradient): 4 (float), or 2 (bf16) in an IVF_MATH_BF16ACT plan -- *ptr then addresses bf16 elements. */
int ivf_i3d_act_elem_bytes(const ivf_i3d_t* net);

/* The hot loop, smth:193-214, for b clips with per-clip masks, entirely on the
 * device: N iterations of sigmoid/L1/TV -> freeze -> forward -> score -> backward
 * -> freeze backward -> Adam.  raw_mask, exp_avg, exp_avg_sq [b,T] in/out;
 * target [b]; traj [N,b,4] = (loss,l1,tv,score) or NULL; first_step = Adam step
 * number of the first iteration (1 for a fresh search); mode = the perturbation the loop
 * optimises through (temporalMaskType, smth:121,202): 0 freeze, 1 reverse. */
int ivf_i3d_search(ivf_i3d_t* net, const float* x, int b, const int* target, float* raw_mask,
                   float* exp_avg, float* exp_avg_sq, float lam1, float lam2, float lr, float beta1,
                   float beta2, float eps, int N, int first_step, int mode, float* traj,
                   ivf_stream_t stream);

/* Scores of perturbed clips (init_mask, mask.py:121-154, and the reverse score,
 * smth:234-235): mode 0 = freeze with mask [b,T] as given (no sigmoid), mode 1 =
 * reverse with mask [b,T]; probs [b,K]. */
int ivf_i3d_perturbed_forward(ivf_i3d_t* net, const float* x, int b, const float* mask, int mode,
                              float* probs, ivf_stream_t stream);

/* GradCamVideo.__call__ for archType "I3D" / target layer Mixed_5c,
 * grad_cam_videos.py:64-142, for b clips.  target[b] (device) selects the class;
 * cam [b,T'*(T/T'),out_h,out_w] (input_spatial_size, grad_cam_videos.py:54-57);
 * probs [b,K] optional. */
int ivf_i3d_gradcam(ivf_i3d_t* net, const float* x, int b, const int* target, int per_frame, int out_h,
                    int out_w, float* cam, float* probs, ivf_stream_t stream);
/* The same for any single endpoint of the model as target layer (pytorch-grad-cam/grad-cam.py:23-54
 * hooks the OUTPUT of the named module: Conv3d_1a_7x7, MaxPool3d_2a_3x3, ..., Mixed_5c): forward,
 * backward-data from the class score down to that endpoint with ITS gradient left ungated, then the
 * Grad-CAM reduction on [T',H',W',C'] and the resize with step = T / T'. */
int ivf_i3d_gradcam_layer(ivf_i3d_t* net, const float* x, int b, const int* target, const char* layer,
                          int per_frame, int out_h, int out_w, float* cam, float* probs, ivf_stream_t stream);
/* argmax over K of probs [b,K] -> target [b] (np.argmax, grad_cam_videos.py:69-70). */
int ivf_argmax(const float* probs, int b, int K, int* target, ivf_stream_t stream);

/* ------------------------------------------------------------------ CLSTM_4 */

typedef struct {
  int B;                 /* maximum clips per call */
  int C, T, H, W;        /* clip geometry, reference NCTHW (KTH: C in {1,3}, 32, 120, 160) */
  int hidden;            /* nb_lstm_units (<= 4) */
  int layers;            /* lstm_layers */
  int kernel;            /* conv_kernel_size[0], odd */
  int stride;            /* conv_stride of the input convolutions */
  int num_classes;
  int softmax;           /* add_softmax (CLSTM_4.py:82-83) */
  int batch_norm;        /* the single shared BatchNorm2d (convolution_lstm.py:85,123) */
  int out_step;          /* step whose pooled top-layer output feeds endFC: the LAST effective
                            step (CLSTM_4.py:78-80 with use_entire_seq=False) */
  /* use_entire_seq=True (CLSTM_4.py:73-76): endFC reads the pooled top-layer outputs of ALL the
   * effective steps reached, concatenated per clip in step order (n_out_steps * feat inputs);
   * n_out_steps == 0 selects the single out_step above. */
  int n_out_steps;
  int out_steps[16];
} ivf_clstm_config;

typedef struct ivf_clstm ivf_clstm_t;

/* models/CLSTM_4.Model + models/convolution_lstm.ConvLSTM (forward :96-132, cell :38-48). */
int ivf_clstm_create(const ivf_clstm_config* cfg, ivf_clstm_t** out);
void ivf_clstm_destroy(ivf_clstm_t* net);
size_t ivf_clstm_weights_bytes(const ivf_clstm_t* net);
size_t ivf_clstm_workspace_bytes(const ivf_clstm_t* net);
int ivf_clstm_bind(ivf_clstm_t* net, void* weights_arena, void* workspace);
/* One cell's reference tensors: Wx* [hid][cin][k][k] + bias [hid], Wh* [hid][hid][k][k];
 * gate order i, f, c, o (convolution_lstm.py:22-29). */
int ivf_clstm_load_cell(ivf_clstm_t* net, int layer, const float* wxi, const float* wxf, const float* wxc,
                        const float* wxo, const float* bxi, const float* bxf, const float* bxc,
                        const float* bxo, const float* whi, const float* whf, const float* whc,
                        const float* who, ivf_stream_t stream);
/* clstm.bn.* (NULL when batch_norm == 0) and endFC.{weight [K][feat], bias}. */
int ivf_clstm_load_head(ivf_clstm_t* net, const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                        const float* bn_var, const float* fc_w, const float* fc_b, float bn_eps,
                        ivf_stream_t stream);
/* Model.forward, CLSTM_4.py:69-85, on b <= B clips (x NCTHW). */
int ivf_clstm_forward(ivf_clstm_t* net, const float* x, int b, float* logits, float* probs,
                      ivf_stream_t stream);
/* BPTT backward-data of the last forward: dx NCTHW [b,C,T,H,W]. */
int ivf_clstm_backward(ivf_clstm_t* net, int b, const int* target, const float* dout, float* score, float* dx,
                       ivf_stream_t stream);
/* The hot loop (KTH:250-270) with the ConvLSTM backbone; arguments as ivf_i3d_search. */
int ivf_clstm_search(ivf_clstm_t* net, const float* x, int b, const int* target, float* raw_mask,
                     float* exp_avg, float* exp_avg_sq, float lam1, float lam2, float lr, float beta1,
                     float beta2, float eps, int N, int first_step, int mode, float* traj,
                     ivf_stream_t stream);
int ivf_clstm_perturbed_forward(ivf_clstm_t* net, const float* x, int b, const float* mask, int mode,
                                float* probs, ivf_stream_t stream);

/* Per-layer kernel selection.  ivf_i3d_autotune times every candidate variant of every
 * convolution (forward and backward-data) on `b` clips and keeps the fastest; the result is
 * 2*ivf_i3d_num_conv_ops() ints that can be read and installed again, e.g. broadcast from
 * rank 0 so all GPUs of a sharded run use identical kernels (bit-identical per-clip results). */
int ivf_i3d_num_conv_ops(const ivf_i3d_t* net);
int ivf_i3d_autotune(ivf_i3d_t* net, int b, int reps, ivf_stream_t stream);
int ivf_i3d_get_tuning(const ivf_i3d_t* net, int* variants_host);
int ivf_i3d_set_tuning(ivf_i3d_t* net, const int* variants_host);

/* Algorithmic forward FLOPs of all Unit3D convolutions for one clip (2*MAC, real
 * channel counts; backward-data costs the same again).  SURVEY.md section 8d. */
double ivf_i3d_conv_flops_per_clip(const ivf_i3d_t* net);

/* ------------------------------------------------------------------ visualisation */

/* create_image_arrays, visualisation.py:96-130 (RESIZE_FLAG = 0 as in both drivers): per frame the strip
 * original | heat-map overlay | perturbed clip, BGR uint8 [T][H][3W][3].  clip, perturbed [3,T,H,W]
 * (RGB planes, 0..255), cam [T,H,W] in [0,1] (GradCamVideo output), lut_bgr [256][3] =
 * cv2.COLORMAP_JET in BGR order (host-provided), frame_max: T floats of workspace.
 * overlay = (lut[uint8(255 cam)] + original) / max over the frame, then uint8(255 .). */
int ivf_viz_blend(const float* clip, const float* cam, const float* perturbed, const unsigned char* lut_bgr,
                  float* frame_max, unsigned char* out, int T, int H, int W, ivf_stream_t stream);
/* vizualize_results_on_gradcam, visualisation.py:35-64: the red (mask 1) / green (mask 0) dot row on the
 * third panel of every frame, 255 for the frame's own dot, 150 for the others; mask_snapped [T] holds 0/1
 * (find_temp_mask_red_dots :67-93 snaps the caller's mask first).  image_width/height: the reference's
 * defaults are 224 whatever the frame size. */
int ivf_viz_dots(unsigned char* img, const float* mask_snapped, int T, int H, int W3, int image_width,
                 int image_height, ivf_stream_t stream);

/* ------------------------------------------------------------------ TF-style ConvLSTM search variant
 *
 * DOCUMENTED EXTENSION, PARITY UNPINNED: the reference's TensorFlow half (video_features_tf/mask/find_mask_kth.py:300-372,
 * 431-452; mask/gradcam.py:28-111; models/clstm.py:9-52, 87-126) runs the temporal-mask search and a per-frame
 * Grad-CAM over a Keras ConvLSTM2D stack.  TF 1.12 / Keras cannot be installed where this library is built, so no
 * output of that code pins these entry points; they follow the published Keras ConvLSTM2D definition as the call site
 * configures it (csrc/tf_clstm.hip has the equations): x-part conv with `padding` 'valid' | 'same' (TensorFlow rule)
 * and `stride`, recurrent conv 'same' / stride 1, kernel kh x kw (any, e.g. the configs' 3 x 5), gate order i, f, c, o,
 * hard-sigmoid recurrent activation (TF 1.12 default), MaxPooling2D(2x2) per frame after every layer, no batch norm
 * (find_mask_kth.py:331), dense head over the flattened (NHWC) last element or whole sequence.
 * Weights in Keras layouts: kernel [kh][kw][Cin][4F], recurrent_kernel [kh][kw][F][4F], bias [4F], dense kernel
 * [inputs][classes].  Clips are NCTHW like everywhere else in this library. */
typedef struct {
  int B;                      /* maximum clips per call */
  int C, T, H, W;             /* clip geometry */
  int layers;                 /* ConvLSTM2D blocks (<= 8) */
  int units[8];               /* filters per block (FLAGS.layers) */
  int kh, kw;                 /* kernel_size_1, kernel_size_2 */
  int stride;                 /* FLAGS.strides */
  int padding;                /* 0 'valid', 1 'same' */
  int recurrent_hard_sigmoid; /* 1: hard_sigmoid (TF 1.12 Keras default), 0: sigmoid */
  int only_last;              /* only_last_element_for_fc == 'yes' */
  int num_classes;
} ivf_tfclstm_config;

typedef struct ivf_tfclstm ivf_tfclstm_t;

int ivf_tfclstm_create(const ivf_tfclstm_config* cfg, ivf_tfclstm_t** out);
void ivf_tfclstm_destroy(ivf_tfclstm_t* net);
size_t ivf_tfclstm_weights_bytes(const ivf_tfclstm_t* net);
size_t ivf_tfclstm_workspace_bytes(const ivf_tfclstm_t* net);
int ivf_tfclstm_bind(ivf_tfclstm_t* net, void* weights_arena, void* workspace);
/* conv output (Ho, Wo), pooled (Hp, Wp) and filters of a block; inputs of the dense head */
int ivf_tfclstm_layer_dims(const ivf_tfclstm_t* net, int layer, int* Ho, int* Wo, int* Hp, int* Wp, int* units);
int ivf_tfclstm_fc_inputs(const ivf_tfclstm_t* net);
int ivf_tfclstm_load_layer(ivf_tfclstm_t* net, int layer, const float* kernel, const float* recurrent_kernel,
                           const float* bias, ivf_stream_t stream);
int ivf_tfclstm_load_head(ivf_tfclstm_t* net, const float* dense_kernel, const float* dense_bias, ivf_stream_t stream);
/* clstm.clstm(x) + softmax (find_mask_kth.py:331-332): logits / probs [b,K] (either may be NULL) */
int ivf_tfclstm_forward(ivf_tfclstm_t* net, const float* x, int b, float* logits, float* probs, ivf_stream_t stream);
/* BPTT of probs[b, target[b]] to the clip: score [b] optional, dx NCTHW */
int ivf_tfclstm_backward(ivf_tfclstm_t* net, int b, const int* target, float* score, float* dx, ivf_stream_t stream);
/* the tf.scan freeze recurrence (find_mask_kth.py:318-327) with mask [b,T] AS GIVEN, then the model: probs [b,K].
 * (The TF graph always applies sigmoid to its mask variable, also in init_mask and in the baseline prediction: callers
 * that mimic it pass sigmoid(values).) */
int ivf_tfclstm_perturbed_forward(ivf_tfclstm_t* net, const float* x, int b, const float* mask, float* probs,
                                  ivf_stream_t stream);
/* N iterations of the search (find_mask_kth.py:356-372, 431-452) on the device; tf.train.AdamOptimizer's update
 * (epsilon beside sqrt(v) before the bias correction); traj [N,b,4] = (loss, l1, tv, score) or NULL */
int ivf_tfclstm_search(ivf_tfclstm_t* net, const float* x, int b, const int* target, float* raw_mask, float* exp_avg,
                       float* exp_avg_sq, float lam1, float lam2, float lr, float beta1, float beta2, float eps, int N,
                       int first_step, float* traj, ivf_stream_t stream);
/* mask/gradcam.py:28-111: per-frame Grad-CAM on the last ConvLSTM2D's output sequence from the class LOGIT;
 * mask [b,T] as given to the freeze recurrence (NULL: unperturbed clip); per_frame 1 = normalization_mode 'frame',
 * 0 = 'sequence'; cam [b,T,out_h,out_w] (bilinear resize restating skimage.transform.resize, unpinned) */
int ivf_tfclstm_gradcam(ivf_tfclstm_t* net, const float* x, int b, const float* mask, const int* target, int per_frame,
                        int out_h, int out_w, float* cam, float* probs, ivf_stream_t stream);

/* ------------------------------------------------------------------ measurement */

/* HIP-event timing of the convolution launches on their own stream, sampled on every
 * `every`-th iteration of ivf_*_search (bench.py's roofline leg).  collect: arrays of
 * IVF_PROFILE_CLASSES entries indexed by kernel variant id (IVF_CONV_IGEMM_BASE + tile for
 * fp32, +3 for split-bf16; IVF_CONV_HALO_BASE + i): summed kernel milliseconds, launch count
 * and algorithmic FLOPs of the sampled launches. */
#define IVF_PROFILE_CLASSES 96
int ivf_profile_enable(int every, int max_launches);
int ivf_profile_disable(void);
int ivf_profile_collect(double* kernel_ms_host, long long* launches_host, double* flops_host);
/* The same sample per launch SITE of an I3D plan (site = 2 * op index + direction; names from
 * ivf_i3d_site_name, count from ivf_i3d_num_sites): summed milliseconds, launches, algorithmic FLOPs and the
 * kernel variant that served the site.  Does not reset the sample: call it before ivf_profile_collect. */
int ivf_profile_collect_sites(double* kernel_ms_host, long long* launches_host, double* flops_host,
                              int* variant_host, int max_sites);
int ivf_i3d_num_sites(const ivf_i3d_t* net);
int ivf_i3d_site_name(const ivf_i3d_t* net, int site, char* name64);
/* Kernel template instance behind a class id ("" until that class has been launched). */
const char* ivf_profile_class_name(int cls);

#ifdef __cplusplus
}
#endif
#endif /* IVF_HIP_H */
