/*
 * checkerpose_hip.h -- C ABI of libcheckerpose_hip.so (MI355X / gfx950).
 *
 * The reference (RuyiLian/CheckerPose) has NO native code and NO FFI: its hot path is the torch-op
 * sequence inside InitNet_GNN.forward / PoseNet_GNNskip.forward.  Each entry point below therefore
 * replaces a span of reference Python (cited per function, paths relative to
 * /root/reference/checkerpose) rather than an existing foreign function.  The binding a maintainer
 * adds on the reference side is the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers + ints, no torch / HIP types in signatures; `cp_stream_t` is a hipStream_t
 *     passed as void* (NULL = default stream).
 *   - every buffer (activations, packed weights, outputs) is caller-owned DEVICE memory; the
 *     library allocates nothing and is re-entrant and stream-ordered.  Its only process state: the thread-local
 *     cp_last_kernel note, and idempotent per-DEVICE caches (an atomic bit per device ordinal = "the large-LDS
 *     function attributes are set on this device", the device's CU count) -- calls are made on the caller's
 *     current device (hipSetDevice), several devices and several threads per process are fine.
 *   - activations are channels-last: (B, H, W, Cphys) with Cphys a multiple of cp_chan_align(dtype);
 *     padded channels are zero.  Graph features are (B, N, Cphys), i.e. the same layout with H=1.
 *   - returns CP_OK (0) or a negative code; cp_strerror() names it.
 */
#ifndef CHECKERPOSE_HIP_H
#define CHECKERPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cp_stream_t;

/* CP_F16 (IEEE half storage, f16 MFMA, fp32 accumulate; values saturate at +-65504): accepted ONLY by the keypoint-side entry
 * points that say so (cp_edgeconv_fused_t, cp_edgeconv_tiled_t, cp_mlp_pair_fused*_t, cp_mlp_query_fused_t, cp_index2feat_conv_t,
 * cp_pack_gemm_weight, cp_conv2d_igemm with CpConvDesc.dtype = CP_F16 / out_f32 = 2) -- the bf16 program runs its per-keypoint
 * (GNN) block group in f16: same bytes and MFMA rate, 3 more mantissa bits where EdgeConv's neighbour differences cancel. */
enum { CP_F32 = 0, CP_BF16 = 1, CP_F16 = 2 };
enum { CP_ACT_NONE = 0, CP_ACT_RELU = 1, CP_ACT_LEAKY = 2 };
enum { CP_LOSS_BCE = 0, CP_LOSS_L1 = 1 };   /* loss_type of losses/code_loss.py ("CE": cp_masked_ce_loss) */
enum {
  CP_OK = 0,
  CP_ERR_INVALID = -1,   /* bad argument / unsupported shape */
  CP_ERR_HIP = -2,       /* a HIP runtime call failed (launch error, bad pointer, ...) */
  CP_ERR_ALIGN = -3,     /* pointer or stride not aligned as the kernel requires */
  CP_ERR_RANGE = -4      /* buffer too large for 32-bit buffer addressing (>= 2 GiB) */
};

int cp_version(void);
const char* cp_strerror(int code);
/* profiling aid: symbol of the HIP kernel the calling thread's most recent entry point launched, spelled as
 * rocprofv3 --kernel-trace prints it (e.g. "conv_igemm_kernel<BF16Tag, 2, 4>"); "" before the first launch. */
const char* cp_last_kernel(void);
/* the same for entry points that issue SEVERAL launches (cp_edgeconv_tiled: key table + gather): cp_kernel_log_begin() clears the
 * calling thread's log, cp_kernel_log() returns every symbol launched since, joined by " + " in launch order (the log is bounded:
 * 1 KB, later symbols are dropped) -- bench.py prices such a call as the SET of its launches. */
void cp_kernel_log_begin(void);
const char* cp_kernel_log(void);

/* Deterministic training mode (process-wide switch; default off).  The reference's step (train.py:303-320) is deterministic on
 * CPU; the default training entry points here accumulate BatchNorm sums, small weight gradients and Index2Feat's scatter with
 * floating-point atomics, whose order varies from run to run.  With the switch on: cp_bn_* accumulate one block per accumulator
 * set (cp_bn_acc_doubles() grows to 64 sets; consumers add the sets in index order), cp_conv2d_wgrad_* always go through partial
 * tiles + the fixed-order reduction (a call without a workspace is refused: CP_ERR_INVALID), cp_index2feat_gather_bwd* sums each
 * patch pixel's contributions in keypoint order.  Same inputs + same launch plan => bit-identical gradients.  The switch is
 * read when an entry point is CALLED (and when a size query is answered): set it before building launch programs / graphs and
 * rebuild them after changing it.  The eval path has no atomics and ignores it. */
void cp_set_deterministic(int on);
int cp_get_deterministic(void);

/* Device-side error channel: one sticky status word per device, OR-ed into by kernels that can detect a failure of their own, read by
 * the host wherever it synchronises anyway (the drop-in models: program build, invalidate(), check_device_status()).  The call
 * synchronises the device (a plain hipMemcpy), creates the word on first use (call it once OUTSIDE a stream capture before launching
 * -- the models do) and, with clear != 0, resets a non-zero word.  Bits:
 *   CP_STATUS_CHAIN0_HANDOVER  the pipelined 64 x 64 chain (cp_hr_branch_chain*, hr_chain0p_kernel) gave up a bounded wait for a row
 *                              hand-over between its waves: that launch's output is WRONG (the bound keeps the box alive, this bit
 *                              keeps the result honest)
 *   CP_STATUS_CHAIN0_STAGING   the same kernel's staging / tail loop ran out of its bound before finishing its rows */
enum { CP_STATUS_CHAIN0_HANDOVER = 1, CP_STATUS_CHAIN0_STAGING = 2 };
int cp_device_status(uint32_t* status_out, int clear);

/* elements per 16 bytes: 4 (f32) or 8 (bf16).  Physical channel counts are multiples of this. */
int cp_chan_align(int dtype);

/* ---------------------------------------------------------------------------------------------
 * Weight packing (init / load_state_dict time, not the hot path).
 * Packs a PyTorch-layout fp32 weight into MFMA-fragment order for cp_conv2d_igemm:
 *   K index = (r, s, cin) with cin padded to cin_phys; blocks of 1 KiB = one (16-channel tile,
 *   K-chunk) wave fragment, ordered [tile][chunk][lane][16 B].
 * transposed = 0 : w is (Cout, Cin, R, S)      (nn.Conv2d / nn.Linear with R=S=1)
 * transposed = 1 : w is (Cin, Cout, 3, 3) of ConvTranspose2d(k3,s2,p1,op1) (pipeline.py:187-197) and
 *                  `phase` = 2*a+b selects the sub-pixel phase (output row parity a, column parity b);
 *                  the packed kernel then has R=1+a, S=1+b taps.
 * row_map (optional, length cout_rows): packed output row n takes source row row_map[n] (or zeros if
 *   row_map[n] < 0); NULL = identity.  Used to interleave zero rows (init MLP, init.py:107).
 * ------------------------------------------------------------------------------------------- */
size_t cp_packed_weight_bytes(int dtype, int cout_rows, int cin_phys, int R, int S);
int cp_pack_conv_weight(cp_stream_t stream, int dtype, const float* w, int Cout, int Cin, int R, int S,
                        int cin_phys, int transposed, int phase, const int32_t* row_map, int cout_rows,
                        void* packed);

/* ---------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on MFMA (fp32: v_mfma_f32_16x16x4_f32, bf16: v_mfma_f32_16x16x32_bf16),
 * fused per-channel affine (folded BatchNorm or bias), optional residual add, activation.
 * Replaces every nn.Conv2d(+BatchNorm2d)(+ReLU/LeakyReLU) and nn.Linear on the path:
 *   backbone convs (timm hrnet/resnet via backbone.py:48), init.py:85-95 (conv1x1), init.py:58-62 +
 *   pipeline.py:49-53 (EdgeConv 1x1 conv, in its factored per-node form), init.py:107 (mlp),
 *   pipeline.py:183-211 (decoder), pipeline.py:146-147 (patch_generator), pipeline.py:61-69,168-180
 *   (MLPs), pipeline.py:349 (seg_block).
 *
 *   out[o_base + b*o_sb + oy*o_sy + ox*o_sx + c*o_sc] =
 *       act( conv(in)[b,oy,ox,c] * scale[c] + shift[c] + residual[same index] )
 * ------------------------------------------------------------------------------------------- */
typedef struct CpConvDesc {
  int32_t dtype;          /* CP_F32 | CP_BF16: type of in / packed weights / residual / out */
  int32_t out_f32;        /* 1: out (and residual) are fp32 regardless of dtype (final logits, seg); 2 (cp_conv2d_igemm, cp_conv2x2_halo,
                             dtype CP_BF16, no residual): out rows are IEEE half (CP_F16: the producer of a keypoint-side tensor) */
  int32_t B, H, W;        /* input spatial size */
  int32_t Cin;            /* channels contracted (physical, multiple of cp_chan_align) */
  int32_t in_cstride;     /* elements between consecutive input pixels (>= in_coff + Cin) */
  int32_t in_coff;        /* first channel of the input slice */
  int32_t R, S, stride, pad;
  int32_t Ho, Wo;         /* output spatial size (iy = oy*stride - pad + r) */
  int32_t Cout;           /* channels stored (physical) ; weights were packed with cout_rows >= Cout */
  int32_t act;            /* CP_ACT_* */
  float slope;            /* LeakyReLU negative slope */
  int32_t ksplit;         /* cp_conv2d_igemm: -1 = never use the split-K variant (results then do not depend on the batch
                             size, bit for bit); anything else = the library decides by shape (cp_conv2d_igemm_splitk).
                             Sits in what was alignment padding: the struct's size and offsets are unchanged. */
  int64_t o_base, o_sb, o_sy, o_sx, o_sc;   /* output (and residual) element strides */
} CpConvDesc;

int cp_conv2d_igemm(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w,
                    const float* scale, const float* shift, const void* residual, void* out);
/* Small-batch routing query: 0, or the number of waves cp_conv2d_igemm would split this conv's K walk over (bf16, long K,
 * a grid that would leave most CUs idle: the split-K variant of the kernel).  M = B*Ho*Wo output pixels, K = R*S*Cin with
 * Cin the PHYSICAL input channels, Cout the physical output channels.  A host that has a specialised large-batch kernel
 * for the layer (halo / row-GEMM) asks this first and, if non-zero, packs the generic weight image instead. */
int cp_conv2d_igemm_splitk(int dtype, long long M, int K, int Cout);

/* ---------------------------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 specialisation with an LDS-staged input halo tile (decoder convs pipeline.py:183-211,
 * HRNet/ResNet body convs): same arithmetic and descriptor as cp_conv2d_igemm (requires R=S=3, stride=1, pad=1,
 * o_sc=1, out_f32=0) but its own packed-weight image ([32-ch group][chunk][tap][tile][lane][16 B], rows permuted
 * for 16-byte epilogue stores).  scale/shift must be readable 8 floats at a time (pad to a multiple of 8).
 * ------------------------------------------------------------------------------------------- */
size_t cp_packed_halo_weight_bytes(int dtype, int Cout, int cin_phys);
int cp_pack_conv3x3_halo_weight(cp_stream_t stream, int dtype, const float* w, int Cout, int Cin, int cin_phys,
                                void* packed);
int cp_conv3x3_halo(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w,
                    const float* scale, const float* shift, const void* residual, void* out);
/* Grouped form for small layers: up to 16 INDEPENDENT 3x3 / stride 1 / pad 1 convs with <= 80 physical output channels on maps of at
 * least 8 x 16 pixels (the branches of an HRNet module at equal depth -- timm HighResolutionModule.forward runs them one after the
 * other; at the training batch each is a ~9 us launch) in ONE launch.  cp_conv3x3_halo_item: cp_conv3x3_halo's arguments, checked
 * and packed on the host (weights packed as for cp_conv3x3_halo); cp_conv3x3_halo_group: items in device memory, prefix = exclusive
 * prefix sum (n_items + 1) of item.blocks, lds_bytes = the largest item.lds_bytes.  Results are bit-identical to the single launches.
 * A residual may alias the output (in-place accumulation) but no two items may write the same tensor. */
/* k = 2 / stride 1 / pad 1 conv with <= 80 output channels off the same LDS-staged halo tile (Index2Feat_module.patch_generator,
 * pipeline.py:144-145,156, where it runs over the whole map: N = 4096 keypoints, the low-resolution stages at N = 512): descriptor as
 * cp_conv2d_igemm with R = S = 2, stride 1, pad 1, Ho = H + 1, Wo = W + 1, o_sc = 1; weights by cp_pack_conv2x2_halo_weight from the
 * fp32 (Cout, Cin, 2, 2) tensor; scale / shift / residual / activation as cp_conv3x3_halo. */
int cp_conv2x2_halo_supported(int dtype, int H, int W, int Cout_phys);
size_t cp_packed_conv2x2_halo_weight_bytes(int dtype, int Cout, int Cin_phys);
int cp_pack_conv2x2_halo_weight(cp_stream_t stream, int dtype, const float* w, int Cout, int Cin, int cin_phys, void* packed);
int cp_conv2x2_halo(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w, const float* scale,
                    const float* shift, const void* residual, void* out);

typedef struct CpConvGroupItem {
  int32_t NT; uint32_t blocks, lds_bytes, pad;
  unsigned long long params[25];       /* opaque: the kernel's parameter block */
} CpConvGroupItem;
int cp_conv3x3_halo_group_supported(int dtype, int H, int W, int Cout_phys);
int cp_conv3x3_halo_item(const CpConvDesc* d, const void* in, const void* packed_w, const float* scale, const float* shift,
                         const void* residual, void* out, CpConvGroupItem* item);
int cp_conv3x3_halo_group(cp_stream_t stream, int dtype, const CpConvGroupItem* items_dev, const uint32_t* prefix_dev, int n_items,
                          uint32_t total_blocks, uint32_t lds_bytes);

/* conv3x3( UpsamplingBilinear2d(scale_factor=2)(in) ) without the upsampled tensor: the decoder's `up_net[1..2]` upsample + first
 * conv (pipeline.py:199-200, get_gdrn_upsample_module).  Descriptor as cp_conv3x3_halo, except that d->H, d->W (= Ho, Wo, both
 * even) are the UPSAMPLED size and `in` is the (B, H/2, W/2, in_cstride) source; Cout a multiple of 256 (the wide kernel), no
 * residual.  Each halo pixel is interpolated (align_corners=True, the arithmetic of cp_upsample2x_bilinear_ac bit for bit) while
 * the tile is staged.  Weights: cp_pack_conv3x3_halo_weight. */
int cp_conv3x3_halo_up2x_supported(int dtype, int Cout);
int cp_conv3x3_halo_up2x(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w,
                         const float* scale, const float* shift, void* out);

/* conv3x3/s1/p1 + folded BN + act (Cout == 256) WITH the 1x1 head that reads its output fused into the epilogue: the decoder's
 * last conv + `seg_block` (pipeline.py:349,383: Conv2d(256 -> S) + bias on the last feature map).  `out` is written as by
 * cp_conv3x3_halo; seg_out (B, S, H, W) fp32 NCHW = seg_b[s] + sum_c seg_w[s][c] * out[b, y, x, c] (the stored, i.e.
 * dtype-rounded, activations; seg_w = [S][256] fp32, pre-rounded to `dtype` by the caller).  S <= 2. */
int cp_conv3x3_halo_seg_supported(int dtype, int Cout, int S);
int cp_conv3x3_halo_seg(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w, const float* scale,
                        const float* shift, void* out, const float* seg_w, const float* seg_b, int S, float* seg_out);

/* 3x3 / stride 2 / pad 1 convolution with a wide input and at most 48 output channels (bf16): HRNet `transition1[1]`
 * (256 -> 36 at 64 x 64 -> 32 x 32; timm HighResolutionNet.transition1 inside backbone.py:48-49).  Descriptor as
 * cp_conv2d_igemm with R = S = 3, stride 2, pad 1, Ho = H / 2, Wo = W / 2; H a multiple of 8, W in {32, 64}, Cin a multiple of
 * 32, d->Cout = physical output channels (multiple of 8, <= 48), scale / shift padded to a multiple of 16 floats.  A
 * workgroup stages 8 input rows of one 32-channel chunk in LDS (every input byte is fetched 9/8 times instead of 2.4). */
int cp_conv3x3_s2_small_supported(int H, int W, int cin_phys, int out_cphys);
size_t cp_conv3x3_s2_small_weight_bytes(int cin_phys, int out_cphys);
int cp_pack_conv3x3_s2_small_weight(cp_stream_t stream, const float* w, int Cout, int Cin, int cin_phys, int out_cphys, void* packed);
int cp_conv3x3_s2_small(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w, const float* scale,
                        const float* shift, void* out);

/* Fused timm BasicBlock of the HRNet branches (C -> C channels, C <= 32 and one 64-byte chunk, stride 1):
 *   out = relu( conv3x3(relu(conv3x3(x)*s1+t1))*s2+t2 + x )     -- intermediate and residual never leave LDS.
 * packed_w1: cp_pack_conv3x3_rows_weight (unpermuted rows); packed_w2: cp_pack_conv3x3_halo_weight.
 * Descriptor as cp_conv3x3_halo with Cin == Cout; `out` must not alias `in`. */
int cp_pack_conv3x3_rows_weight(cp_stream_t stream, int dtype, const float* w, int Cout, int Cin, int cin_phys,
                                void* packed);
int cp_basicblock_fused(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w1,
                        const float* scale1, const float* shift1, const void* packed_w2, const float* scale2,
                        const float* shift2, void* out);

/* timm resnet.Bottleneck of HRNet layer1 (inside timm.create_model, backbone.py:35) in ONE launch:
 *   out = relu( bn3(conv1x1( relu(bn2(conv3x3( relu(bn1(conv1x1(x))) ))) )) + shortcut(x) ),  Cin -> 64 -> 64 -> 256
 *   blocks 1..3: Cin = 256, shortcut = identity (packed_wd = scaled = shiftd = NULL);
 *   block 0    : Cin = 64,  shortcut = bn_d(conv1x1_d(x)) (`downsample`), packed_wd (256,64,1,1) + its folded BN.
 * bf16 storage only (the x halo tile, both intermediates and the output tile live in <= 140 KB of LDS).  d: dtype
 * CP_BF16, Cin as above, Cout = 256, stride 1, H/W/B, input slice and output strides as for cp_conv2d_igemm;
 * in != out.  All weights in the generic image of cp_pack_conv_weight: w1 (64,Cin,1,1) cin_phys Cin; w2 (64,64,3,3)
 * cin_phys 64; w3 / wd (256,64,1,1) cin_phys 64.  scale/shift: folded BatchNorm (64, 64, 256, 256 floats, 16-byte
 * aligned). */
int cp_bottleneck_fused(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w1,
                        const float* scale1, const float* shift1, const void* packed_w2, const float* scale2,
                        const float* shift2, const void* packed_w3, const float* scale3, const float* shift3,
                        const void* packed_wd, const float* scaled, const float* shiftd, void* out);

/* ---------------------------------------------------------------------------------------------
 * 1x1 conv / Linear specialisation with LDS-staged rows (EdgeConv node GEMMs, 256-wide MLPs pipeline.py:61-69,
 * 168-180, conv1x1 init.py:85-95, incre conv3): same arithmetic and descriptor as cp_conv2d_igemm (requires
 * R=S=1, stride=1, pad=0, o_sc=1, out_f32=0), own packed image ([32-ch group][chunk][tile][lane][16 B]).
 * ------------------------------------------------------------------------------------------- */
size_t cp_packed_gemm_weight_bytes(int dtype, int Cout, int cin_phys);
int cp_pack_gemm_weight(cp_stream_t stream, int dtype, const float* w, int Cout, int Cin, int cin_phys, void* packed);
int cp_gemm_rows(cp_stream_t stream, const CpConvDesc* d, const void* in, const void* packed_w,
                 const float* scale, const float* shift, const void* residual, void* out);

/* MLP_QueryNet.forward (pipeline.py:168-180: Linear(256 -> 256) + LeakyReLU, Linear(256 -> 64) + LeakyReLU, Linear(64 -> 2); the
 * `pts` argument is unused there) as ONE launch over the B * N keypoint rows, bf16: rows read once, both hidden layers stay on chip
 * (LDS / registers), only the two logits per row leave.  in (B, N, in_cstride) bf16 channels [in_coff, in_coff + 256);
 * packed_w1 / packed_w2: cp_pack_gemm_weight images of the (256, 256) and (64, 256) weights, scale / shift fp32 per output channel
 * (nn.Linear: scale 1, shift = bias); w3 fp32 (2, 64) row-major as nn.Linear stores it, b3 fp32 (2).
 * out fp32: logit c of row (b, n) at out[o_base + b * o_sb + n * o_sn + c * o_sc] (the (B, 13, N) logit block: pipeline.py:375-378). */
int cp_mlp_query_fused_supported(int C0, int C1, int C2, int C3);
int cp_mlp_query_fused(cp_stream_t stream, const void* in, int in_cstride, int in_coff, int B, int N,
                       const void* packed_w1, const float* scale1, const float* shift1, float slope1,
                       const void* packed_w2, const float* scale2, const float* shift2, float slope2,
                       const float* w3, const float* b3, float* out, long long o_base, long long o_sb, long long o_sn, long long o_sc);
/* the same with the rows, both packed weight images (cp_pack_gemm_weight(dtype, ...)) and the on-chip hidden rows in `dtype` = CP_BF16 or CP_F16 */
int cp_mlp_query_fused_t(cp_stream_t stream, int dtype, const void* in, int in_cstride, int in_coff, int B, int N,
                         const void* packed_w1, const float* scale1, const float* shift1, float slope1,
                         const void* packed_w2, const float* scale2, const float* shift2, float slope2,
                         const float* w3, const float* b3, float* out, long long o_base, long long o_sb, long long o_sn, long long o_sc);

/* Refine_moduleGNN.pre_graph_module (pipeline.py:237-240, applied at :283-286: Linear(Cin -> 256) + LeakyReLU, Linear(256 -> 256) +
 * LeakyReLU over the concatenated [local | previous graph] feature rows) as ONE launch, bf16: the hidden rows stay in LDS.
 * in (B, N, in_cstride) channels [in_coff, in_coff + Cin), Cin a multiple of 32, <= 512; packed_w1 / packed_w2: cp_pack_gemm_weight
 * images of the (256, Cin) and (256, 256) weights; bias fp32 (256) each; out (B, N, out_cstride) channels [out_coff, out_coff + 256). */
int cp_mlp_pair_fused_supported(int Cin, int C1, int C2);
int cp_mlp_pair_fused(cp_stream_t stream, const void* in, int in_cstride, int in_coff, int Cin, int B, int N,
                      const void* packed_w1, const float* bias1, float slope1, const void* packed_w2, const float* bias2, float slope2,
                      void* out, int out_cstride, int out_coff);
/* ... in `dtype` = CP_BF16 or CP_F16 (input rows, weights, hidden rows, output rows) */
int cp_mlp_pair_fused_t(cp_stream_t stream, int dtype, const void* in, int in_cstride, int in_coff, int Cin, int B, int N,
                        const void* packed_w1, const float* bias1, float slope1, const void* packed_w2, const float* bias2,
                        float slope2, void* out, int out_cstride, int out_coff);

/* The same with Index2Feat_module's gather (pipeline.py:156-163: four 64-channel taps of patch_generator's output map at
 * (2v, 2u), (2v + k, 2u), (2v, 2u + k), (2v + k, 2u + k), times the {0, 1} RoI bit, :280) done by the kernel's DMA loader: the
 * (B, N, 256) local-feature tensor is never written or read back.  Row (b, n) = [tap 0 | tap 1 | tap 2 | tap 3 | gin row], i.e. the
 * weight packed_w1 is the (256, 256 + Cg) one of cp_mlp_pair_fused.  patches (B, Hp, Wp, p_cstride) bf16, channels
 * [p_coff, p_coff + 64); x_id / y_id int32 (B, N), mask fp32 (B, N) as cp_bits_decode leaves them; zeros: >= 128 zero bytes (the
 * source of rows whose RoI bit is 0); gin (B, N, gin_cstride) channels [gin_coff, gin_coff + Cg), Cg in {64, 128, 192, 256}. */
typedef struct {
  const void* patches;
  const int32_t* x_id;
  const int32_t* y_id;
  const float* mask;
  const void* zeros;
  int32_t p_cstride, p_coff, Hp, Wp, k;
} CpI2fGather;
int cp_mlp_pair_fused_gather_supported(int Cg, int E_ch, int k);
int cp_mlp_pair_fused_gather(cp_stream_t stream, const CpI2fGather* g, const void* gin, int gin_cstride, int gin_coff, int Cg, int B, int N,
                             const void* packed_w1, const float* bias1, float slope1, const void* packed_w2, const float* bias2,
                             float slope2, void* out, int out_cstride, int out_coff);
/* ... in `dtype` = CP_BF16 or CP_F16 (patch map, graph rows, weights, hidden rows, output rows: ONE type) */
int cp_mlp_pair_fused_gather_t(cp_stream_t stream, int dtype, const CpI2fGather* g, const void* gin, int gin_cstride, int gin_coff, int Cg,
                               int B, int N, const void* packed_w1, const float* bias1, float slope1, const void* packed_w2,
                               const float* bias2, float slope2, void* out, int out_cstride, int out_coff);

/* nn.UpsamplingBilinear2d(scale_factor=2) == interpolate(align_corners=True), pipeline.py:199.
 * Reads channels [in_coff, in_coff+C) of (B,H,W,in_cstride), writes [out_coff, ..) of (B,2H,2W,out_cstride). */
int cp_upsample2x_bilinear_ac(cp_stream_t stream, int dtype, const void* in, void* out, int B, int H, int W,
                              int C, int in_cstride, int in_coff, int out_cstride, int out_coff);

/* HRNet fuse layer sum (timm HighResolutionModule.forward): out = relu?( sum_t nearest_up(src_t, 2^shift_t) ),
 * sources (B, H>>shift_t, W>>shift_t, C) contiguous, nsrc <= 4; written to channels [out_coff, out_coff + C) of the
 * (B, H, W, out_cstride) tensor `out` (out_cstride = C, out_coff = 0: a plain tensor). */
int cp_fuse_sum_act(cp_stream_t stream, int dtype, int nsrc, const void* const* srcs, const int32_t* shifts,
                    void* out, int B, int H, int W, int C, int relu, int out_cstride, int out_coff);

/* max_pool2d(3, 2, 1) for the resnet34 stem (timm resnet). (B,H,W,C) -> (B,H/2,W/2,C) */
int cp_maxpool3x3s2(cp_stream_t stream, int dtype, const void* in, void* out, int B, int H, int W, int C);

/* ---------------------------------------------------------------------------------------------
 * EdgeConv aggregation, factored form of StaticGraph_module.forward (init.py:64-68 == pipeline.py:55-59,
 * get_graph_feature init.py:36-49).  pq is the per-node GEMM output (B, N, 2*C): [P' | Q'] with the
 * BatchNorm scale folded in (P' = s*W1 x, Q' = s*(W2-W1) x + t), so
 *   out[b,i,c] = leaky( max_k P'[b, idx[g,i,k], c] + Q'[b,i,c] ),   g = graph_ids ? graph_ids[b] : 0
 * idx is (G, N, K) int32 (LM: per-object tables, pipeline_lm.py:57).  Writes channels
 * [out_coff, out_coff+C) of (B, N, out_cstride).
 * ------------------------------------------------------------------------------------------- */
int cp_edgeconv_gather_max(cp_stream_t stream, int dtype, const void* pq, const int32_t* idx,
                           const int32_t* graph_ids, void* out, int B, int N, int K, int C, int G,
                           int out_cstride, int out_coff, float slope);

/* Index2Feat_module.forward gather (pipeline.py:156-163) fused with the RoI mask multiply (pipeline.py:280):
 * patches (B, Hp, Wp, E); for keypoint (b,i) taps (2y,2x), (2y+k,2x), (2y,2x+k), (2y+k,2x+k) ->
 * out[b,i, out_coff + t*E + e] = patches[b,ty,tx,e] * mask[b,i].  ids int32 (B,N), mask fp32 (B,N). */
int cp_index2feat_gather(cp_stream_t stream, int dtype, const void* patches, const int32_t* x_id,
                         const int32_t* y_id, const float* mask, void* out, int B, int N, int Hp, int Wp,
                         int E, int k, int out_cstride, int out_coff);

/* HRNet stem in one launch (bf16): NCHW fp32 image (B, 3, Hin, Win) -> conv1 3x3/s2 (3 -> 64) + BN + ReLU -> conv2 3x3/s2
 * (64 -> 64) + BN + ReLU -> (B, Hin/4, Win/4, 64) bf16 channels-last (timm hrnet conv1/bn1/conv2/bn2 behind reference
 * backbone.py:48-49); the 2 MB-per-crop intermediate never leaves LDS.  Hin % 32 == 0, Win % 64 == 0.  Weights: fp32
 * (64, 3, 3, 3) and (64, 64, 3, 3) packed by cp_pack_hr_stem_weights into buffers of cp_hr_stem_weight_bytes(0 | 1) bytes;
 * scale / shift = folded BatchNorm, 64 floats each. */
size_t cp_hr_stem_weight_bytes(int which);
int cp_pack_hr_stem_weights(cp_stream_t stream, const float* w1, const float* w2, void* packed1, void* packed2);
int cp_hr_stem(cp_stream_t stream, const float* img_nchw, int B, int Hin, int Win, const void* packed1, const float* scale1,
               const float* shift1, const void* packed2, const float* scale2, const float* shift2, void* out);

/* ---------------------------------------------------------------------------------------------
 * One launch per HRNet branch chain (bf16): the four BasicBlocks of `HighResolutionModule.branches[j]` (inside
 * timm.create_model("hrnet_w18", features_only=True), reference backbone.py:48-49; restated oracle/checkerpose_oracle.py
 * _hr_module) preceded by the previous module's fuse sum, with the crop's whole branch map resident in LDS.
 *   x   = [relu]( sum_k upsample_nearest(src_k, 2^shift_k) )        srcs (B, H>>shift, W>>shift, Cphys) bf16, Cphys = ceil8(C)
 *   for blk in 0..3:  x = relu( bn2(conv3x3(relu(bn1(conv3x3(x))))) + x )
 *   out = x                                                           (B, H, W, Cphys) bf16, pad channels exactly zero
 * Supported (C, H, W): (18, 64, 64), (36, 32, 32), (72, 16, 16), (144, 8, 8) -- the four HRNet-W18 branches of a 256 x 256
 * crop (cp_hr_chain_supported).  The 64 x 64 branch keeps its BasicBlock residuals in `out` between blocks (its map alone fills
 * the LDS), the others keep them on chip.  Weights: cp_pack_hr_chain_weight() packs conv `conv_index` (0..7 = block.conv1, block.conv2, ...)
 * of fp32 (C, C, 3, 3) weights TIMES the folded-BN `scale` of their output channel (fp32 [C]; NULL = 1) into the caller-owned
 * blob of cp_hr_chain_weight_bytes() bytes; `affine` = fp32 [8][2][cp_hr_chain_affine_floats()] per conv, zero beyond C: row 1
 * is the folded-BN shift the accumulators start from, row 0 is NOT read (the scale lives in the weights: a block's epilogue
 * is residual + ReLU + rounding only).  out must not alias a source. */
int cp_hr_chain_supported(int C, int H, int W);
size_t cp_hr_chain_weight_bytes(int C, int H, int W);
int cp_hr_chain_affine_floats(int C, int H, int W);
int cp_pack_hr_chain_weight(cp_stream_t stream, const float* w, const float* scale, int C, int H, int W, int conv_index, void* blob);
int cp_hr_branch_chain(cp_stream_t stream, int B, int C, int H, int W, int nsrc, const void* const* srcs,
                       const int32_t* shifts, int relu_in, const void* packed_w, const float* affine, void* out);

/* cp_hr_branch_chain of the 64 x 64 x 18 branch WITH the stride-2 fuse-layer convs that read its output (kind 1 of cp_hr_fuse_out
 * below: fuse_layers[i][0][0], i = 1..3, 3x3 / stride 2 / pad 1 + folded BN (+ ReLU where the down-sampling chain goes on)) computed
 * off the finished map while it is still in LDS: the separate launch behind the module's longest chain and its re-read of the map
 * disappear.  Up to 3 convs whose padded channel counts (out_cphys, multiples of 8) add up to <= cp_hr_chain_tail_channels() (96:
 * HRNet-W18 has 40 + 24 + 24); conv i owns the 16-byte channel pieces [first_piece_i, first_piece_i + out_cphys_i / 8) with
 * first_piece_0 = 0 and first_piece_{i+1} = first_piece_i + out_cphys_i / 8.
 *   packed_w: one ZERO-FILLED blob of cp_hr_chain_tail_weight_bytes() bytes into which cp_pack_hr_chain_tail_weight() has written every
 *             conv (fp32 (Cout, 18, 3, 3) weights TIMES the folded-BN scale of their output channel);
 *   shift:    fp32 [96], the folded-BN shift at combined channel 8 * first_piece_i + c, zero elsewhere;
 *   out[i]:   (B, 32, 32, out_cphys[i]) bf16, channels [Cout, out_cphys) exactly zero.
 * Same arithmetic as the chain's own convs (bf16 operands with the scale folded in, fp32 accumulate starting from the shift). */
typedef struct CpChainTail {
  const void* packed_w;
  const float* shift;
  void* out[3];
  int32_t nconv;
  int32_t Cout[3], out_cphys[3], relu[3];
} CpChainTail;
int cp_hr_chain_tail_supported(int C, int H, int W);
size_t cp_hr_chain_tail_weight_bytes(void);
int cp_hr_chain_tail_channels(void);
int cp_pack_hr_chain_tail_weight(cp_stream_t stream, const float* w, const float* scale, int Cout, int first_piece, int out_cphys, void* blob);
int cp_hr_branch_chain_tail(cp_stream_t stream, int B, int C, int H, int W, int nsrc, const void* const* srcs,
                            const int32_t* shifts, int relu_in, const void* packed_w, const float* affine, void* out,
                            const CpChainTail* tail);

/* The same for the 36 / 72 / 144-channel chains (round 5): ANY first-level fuse-layer conv that reads the branch -- kind 0: the 1x1
 * conv + BN towards a higher-resolution branch (output at the branch's resolution), kind 1: the first 3x3 / stride-2 conv + BN
 * (+ ReLU when the chain goes on) towards a lower-resolution one (output at half resolution) -- runs in the chain launch's tail off
 * the map in LDS: the grouped cp_hr_fuse_out launch behind the chain and its re-read of the map go away.  Up to 3 convs per launch;
 * weights packed by cp_pack_hr_chain_tailconv_weight (folded-BN scale inside), shift: fp32 [16 ceil(Cout / 16)] (zero beyond Cout);
 * out (B, H >> kind, W >> kind, out_cphys) bf16, out_cphys a multiple of 8 in [Cout, 16 ceil(Cout / 16)].  Supported (kind, Cout)
 * pairs are HRNet-W18's: C = 36: 1x1 -> 18, s2 -> 72 / 36; C = 72: 1x1 -> 18 / 36, s2 -> 144; C = 144: 1x1 -> 18 / 36 / 72. */
typedef struct {
  const void* packed_w;
  const float* shift;
  void* out;
  int32_t kind, Cout, out_cphys, relu;
} CpChainTailConv;
int cp_hr_chain_tailconv_supported(int C, int H, int W, int kind, int Cout);
size_t cp_hr_chain_tailconv_weight_bytes(int C, int H, int W, int kind, int Cout);
int cp_pack_hr_chain_tailconv_weight(cp_stream_t stream, const float* w, const float* scale, int C, int H, int W, int kind, int Cout,
                                     void* packed);
int cp_hr_branch_chain_tails(cp_stream_t stream, int B, int C, int H, int W, int nsrc, const void* const* srcs, const int32_t* shifts,
                             int relu_in, const void* packed_w, const float* affine, void* out, int ntail, const CpChainTailConv* convs);

/* The first-level fuse-layer convs of a timm HighResolutionModule that read ONE branch's output `src` (B, H, W, cin_phys) bf16
 * (timm HighResolutionModule.fuse_layers inside backbone.py:35): up to 4 convs per launch, each
 *   kind 0: 1x1 conv + folded BN at the source resolution (the term towards a higher-resolution branch, before its nearest
 *           upsample), or
 *   kind 1: 3x3 / stride 2 / pad 1 conv + folded BN (+ ReLU when `relu`: a chain that goes on) at half the resolution,
 * out_i = act(conv_i(src) * scale + shift) as (B, Ho, Wo, out_cphys) bf16, channels [Cout, out_cphys) exactly zero.  A
 * workgroup stages a band of src in LDS once and runs every conv off it.  packed_w: cp_pack_hr_fuse_out_weight of the fp32
 * (Cout, Cin, k, k) weight; affine: fp32 [2][cp_hr_fuse_out_affine_floats(out_cphys)] = scale then shift, zero beyond Cout. */
typedef struct CpFuseConv {
  const void* packed_w;
  const float* affine;
  void* out;
  int32_t kind, Cout, out_cphys, relu;
} CpFuseConv;
int cp_hr_fuse_out_supported(int H, int W, int cin_phys);
size_t cp_hr_fuse_out_weight_bytes(int cin_phys, int out_cphys, int kind);
int cp_hr_fuse_out_affine_floats(int out_cphys);   /* 0: that many (padded) output channels are not supported (> 160) */
int cp_pack_hr_fuse_out_weight(cp_stream_t stream, const float* w, int Cout, int Cin, int cin_phys, int out_cphys, int kind,
                               void* packed);
int cp_hr_fuse_out(cp_stream_t stream, const void* src, int B, int H, int W, int cin_phys, int nconv, const CpFuseConv* convs);

/* EdgeConv layer in ONE launch for N = 512 keypoints (bf16): per-node GEMM [P' | Q'] = x . wpq^T on MFMA, P' table of one
 * 64-channel slice in LDS, neighbour gather-max out of LDS, + Q', LeakyReLU (StaticGraph_module, init.py:54-68 ==
 * pipeline.py:45-59; per-sample graphs `knn_idx[obj_ids-1]` of pipeline_lm.py:55-57 through graph_ids).
 *   x (B, 512, in_cstride) channels [in_coff, in_coff + Cin);  wpq fp32 (2 Cout, Cin) = [W1 ; W2 - W1], packed by
 *   cp_pack_edgeconv_fused_weight;  scale / shift fp32 [2 Cout] = [s | s], [0 | t] (folded BatchNorm);  idx int32 (G, 512, K);
 *   out (B, 512, out_cstride) channels [out_coff, out_coff + Cout) = leaky(max_k (s W1 x)_{idx[k]} + s (W2 - W1) x + t).
 * Supported: N = 512, K <= 20 and a multiple of 4, Cin in {64, 256}, Cout in {64, 128, 192, 256} (cp_edgeconv_fused_supported);
 * other shapes (N = 4096) use cp_conv2d_igemm / cp_gemm_rows + cp_edgeconv_gather_max. */
int cp_edgeconv_fused_supported(int N, int K, int Cin, int Cout);
size_t cp_edgeconv_fused_weight_bytes(int Cin, int Cout);
int cp_pack_edgeconv_fused_weight(cp_stream_t stream, const float* wpq, int Cin, int Cout, void* packed);
int cp_edgeconv_fused(cp_stream_t stream, const void* x, int in_cstride, int in_coff, const void* packed_w,
                      const float* scale, const float* shift, const int32_t* idx, const int32_t* graph_ids, void* out,
                      int out_cstride, int out_coff, int B, int N, int K, int Cin, int Cout, int G, float slope);
/* ... in `dtype` = CP_BF16 or CP_F16 (x rows, packed weights, output rows; the gather keys are IEEE halves either way) */
int cp_pack_edgeconv_fused_weight_t(cp_stream_t stream, int dtype, const float* wpq, int Cin, int Cout, void* packed);
int cp_edgeconv_fused_t(cp_stream_t stream, int dtype, const void* x, int in_cstride, int in_coff, const void* packed_w,
                        const float* scale, const float* shift, const int32_t* idx, const int32_t* graph_ids, void* out,
                        int out_cstride, int out_coff, int B, int N, int K, int Cin, int Cout, int G, float slope);

/* EdgeConv layer for LARGE graphs (N = 1024 .. 32768 keypoints in patches of 512, bf16; BASELINE config #5: npt = 4096), the
 * LDS-staged gather of cp_edgeconv_fused, tiled (edgeconv_tiled.hip; same reference lines as above).  The caller works in an
 * INTERNAL keypoint numbering in which every patch of 512 consecutive rows is spatially compact (checkerpose_amd/graph_sched.py:
 * tile_schedule) and passes, per graph g and patch t:
 *   halo int32 (G, N/512, HPAD): internal row ids of the patch's out-of-patch neighbour rows (padded with any valid row);
 *   nbr  int16 (G, N/512, 512, K): every own row's K neighbours as table SLOTS (own row j -> j, halo entry h -> 512 + h).
 * Two launches: the key table P' = s W1 x of every row (written to `key_table`, cp_edgeconv_tiled_table_bytes, plane-major
 * [crop][Cout / 8][N][16 B] IEEE half-precision keys, clamped to +-65504), then per (crop, patch) the table slice in LDS by LDS-DMA, gather-max,
 * Q' on MFMA, LeakyReLU.  x / out / scale / shift as cp_edgeconv_fused; packed_w_fused = cp_pack_edgeconv_fused_weight's image,
 * packed_w_q = cp_pack_edgeconv_tiled_weight's (Q halves in 32-channel slices).
 * Supported (cp_edgeconv_tiled_supported): N a multiple of 512 above 512, K <= 20 and a multiple of 4, Cin in {64, 256},
 * Cout in {64 .. 256 step 64}, HPAD a multiple of 64 with the table in 160 KB of LDS (HPAD <= 1024 at Cin = 256). */
int cp_edgeconv_tiled_supported(int N, int K, int Cin, int Cout, int HPAD);
size_t cp_edgeconv_tiled_weight_bytes(int Cin, int Cout);
size_t cp_edgeconv_tiled_table_bytes(int B, int N, int Cout);
int cp_pack_edgeconv_tiled_weight(cp_stream_t stream, const float* wpq, int Cin, int Cout, void* packed);
int cp_edgeconv_tiled(cp_stream_t stream, const void* x, int in_cstride, int in_coff, const void* packed_w_fused,
                      const void* packed_w_q, const float* scale, const float* shift, const int32_t* halo,
                      const int16_t* nbr, const int32_t* graph_ids, void* key_table, void* out, int out_cstride, int out_coff,
                      int B, int N, int K, int Cin, int Cout, int G, int HPAD, float slope);
/* ... in `dtype` = CP_BF16 or CP_F16 (x rows, both packed weight images, output rows; the key table holds IEEE halves either way) */
int cp_pack_edgeconv_tiled_weight_t(cp_stream_t stream, int dtype, const float* wpq, int Cin, int Cout, void* packed);
int cp_edgeconv_tiled_t(cp_stream_t stream, int dtype, const void* x, int in_cstride, int in_coff, const void* packed_w_fused,
                        const void* packed_w_q, const float* scale, const float* shift, const int32_t* halo, const int16_t* nbr,
                        const int32_t* graph_ids, void* key_table, void* out, int out_cstride, int out_coff, int B, int N, int K,
                        int Cin, int Cout, int G, int HPAD, float slope);
/* The renumbering at the launch program's boundary (perm int32 (G, N): internal row i = original keypoint perm[g][i];
 * graph_ids (B) or NULL).  Rows of `row_bytes` (a multiple of 16): out[b][i] = in[b][perm[g_b][i]].  Columns of (B, R, N)
 * arrays of 4- / 8-byte elements (logit block, ids): scatter = 1: out[b][r][perm[g_b][i]] = in[b][r][i] (internal -> original),
 * scatter = 0: out[b][r][i] = in[b][r][perm[g_b][i]].  Not in place (in == out: CP_ERR_INVALID). */
int cp_permute_rows(cp_stream_t stream, const void* in, void* out, const int32_t* perm, const int32_t* graph_ids, int B, int N,
                    int row_bytes);
int cp_permute_cols(cp_stream_t stream, const void* in, void* out, const int32_t* perm, const int32_t* graph_ids, int B, int R,
                    int N, int elem_bytes, int scatter);

/* Index2Feat_module.forward + RoI mask in one launch (bf16): the patch_generator conv (Conv2d(256 -> 64, k = 2, pad = 1),
 * pipeline.py:146-147) evaluated ONLY at the 4 gathered taps of every keypoint (pipeline.py:156-163), times mask (pipeline.py:280):
 *   out[b, n, 64 t + c] = mask[b, n] * (bias[c] + sum_{dy, dx, ci} w[c, ci, dy, dx] * f[b, py - 1 + dy, px - 1 + dx, ci]),
 *   (py, px) = (2 y_id + k [t & 1], 2 x_id + k [t >> 1]),  t = 0..3 in the reference's sf1..sf4 order.
 * f (B, H, W, in_cstride) channels [in_coff, +256); w fp32 (64, 256, 2, 2) packed by cp_pack_index2feat_conv_weight into
 * cp_index2feat_conv_weight_bytes() bytes; out (B, N, out_cstride) channels [out_coff, +256).  Used where it is cheaper than
 * cp_conv2d_igemm + cp_index2feat_gather (the 64 x 64 stage at N = 512: half the FLOPs, no (H+1) x (W+1) x 64 tensor). */
int cp_index2feat_conv_supported(int Cin, int E_ch, int k);
size_t cp_index2feat_conv_weight_bytes(void);
int cp_pack_index2feat_conv_weight(cp_stream_t stream, const float* w, void* packed);
int cp_index2feat_conv(cp_stream_t stream, const void* f, int in_cstride, int in_coff, const void* packed_w, const float* bias,
                       const int32_t* x_id, const int32_t* y_id, const float* mask, void* out, int B, int N, int H, int W, int k,
                       int out_cstride, int out_coff);
/* ... the (bf16) conv's output rows written as `out_dtype` = CP_BF16 or CP_F16 */
int cp_index2feat_conv_t(cp_stream_t stream, int out_dtype, const void* f, int in_cstride, int in_coff, const void* packed_w,
                         const float* bias, const int32_t* x_id, const int32_t* y_id, const float* mask, void* out, int B, int N,
                         int H, int W, int k, int out_cstride, int out_coff);

/* Bit decode (pipeline.py:72-127, 367-369, 380-381) on the fp32 logit block `bits` (B, 13, N):
 * row 0 = roi, rows 1..6 = x bits (MSB first), rows 7..12 = y bits.
 *   stage < 0 : mask = bit(bits[0]) ; x_id = MSB-first int of rows 1..3 ; y_id of rows 7..9
 *   stage = i : x_id = 2*x_id + bit(bits[4+i]) ; y_id = 2*y_id + bit(bits[10+i])
 * bit(z) = [sigmoid(z) > 0.5 in fp32] = [z > CP_SIGMOID_HALF_Z0]: fp32 sigmoid is exactly 0.5 on 0 <= z <= 1.5 * 2^-24
 * (measured through the reference's from_mask_prob_to_mask; fixture tests/golden/sigmoid_threshold.npz), so `z > 0`
 * would differ from the reference there.  Also mirrors the ids to int64 (the reference's return dtype). */
#define CP_SIGMOID_HALF_Z0_BITS 0x33C00000u   /* largest fp32 z with sigmoidf(z) == 0.5f: 8.940696716e-08 */
int cp_bits_decode(cp_stream_t stream, const float* bits, int stage, float* mask, int32_t* x_id,
                   int32_t* y_id, int64_t* x_id64, int64_t* y_id64, int B, int N);

/* Post-forward decode on the device (next-row N2; reference test.py:294-329 + test_network_with_test_data.py:50-66):
 * bits (B,13,N) fp32 logits, seg (B,2,H,W) fp32 logits (0 = visible, 1 = full), ids int64 (B,N), roi_xy_ori (B,2,H,W)
 * fp32 -> p2d (B,N,2) fp32, valid (B,N,3) uint8 [all | full-mask | visible-mask], count (B,3) int32.
 * discard_bd_pixel = from_id_to_pose's argument of that name (:60-63): d > 0 drops keypoints whose pixel lies within d
 * pixels of the RoI border (d <= x < W-d, d <= y < H-d must hold); 0 = off (the reference's default). */
int cp_correspondences(cp_stream_t stream, const float* bits, const float* seg, const int64_t* x_id, const int64_t* y_id,
                       const float* roi_xy_ori, float* p2d, uint8_t* valid, int32_t* count, int B, int N, int H, int W,
                       int discard_bd_pixel);
/* The same with the coordinate grid built on the fly from the crop's final box (device, B x 4 int32: x, y, w, h = get_final_Bbox,
 * bop_dataset_pytorch.py:188-222) instead of a (B,2,H,W) roi_xy_ori tensor uploaded per batch: p2d = (float)(w / W * x_id + x) etc.
 * in fp64 then fp32, exactly the entries of the loader's grid (mapping_pixel_position_to_original_position_2d :223-235, :380). */
int cp_correspondences_bbox(cp_stream_t stream, const float* bits, const float* seg, const int64_t* x_id, const int64_t* y_id,
                            const int32_t* final_bbox, float* p2d, uint8_t* valid, int32_t* count, int B, int N, int H, int W,
                            int discard_bd_pixel);

/* Pose from the correspondences, on the device (next-row N4; reference test_network_with_test_data.py:100-114: the call
 *   cv2.solvePnPRansac(valid_p3d, valid_disc_p2d, cam_K, None, reprojectionError, iterationsCount, flags=cv2.SOLVEPNP_EPNP)
 * and the identity-pose fallback below 4 valid correspondences), one workgroup per crop, fp64 arithmetic:
 *   p3d fp32 (N,3) model keypoints in their ORIGINAL units (batch stride p3d_bstride elements; 0 = one object for all crops);
 *   p2d fp32 (B,N,2) and valid uint8 with `valid_stride` bytes between keypoints = cp_correspondences' outputs (valid + column c,
 *   stride 3: c = 0 all | 1 in full mask | 2 in visible mask);  cam_K fp32 row-major 3x3 (batch stride K_bstride; 0 = shared);
 *   RANSAC over `iterations` (<= 256) EPnP hypotheses of 5 correspondences drawn by a counter-based
 *   hash of (seed, crop, hypothesis), inlier test squared reprojection error <= reproj_threshold^2, the hypothesis with the most
 *   inliers (>= the sample size; first on ties), final EPnP over its inliers.  Exactly 4 valid correspondences: no RANSAC, as in
 *   OpenCV (model_points == npoints) -- P3P on the first three, the fourth picks among the up-to-four poses, all four are inliers.
 * Outputs: pose fp64 (B,12) = [R row-major | t], inliers uint8 (B,N), status int32 (B): 1 = solved, 0 = identity fallback.
 * scratch: cp_pnp_ransac_scratch_bytes(B, N).  opencv-python is not part of the reference's tree: the algorithm is restated from
 * the publication / OpenCV's structure (oracle/pnp_oracle.py lists the deliberate differences); parity with cv2 is UNPINNED. */
size_t cp_pnp_ransac_scratch_bytes(int B, int N);
int cp_pnp_ransac(cp_stream_t stream, const float* p3d, long long p3d_bstride, const float* p2d, const uint8_t* valid,
                  int valid_stride, const float* cam_K, long long K_bstride, int B, int N, float reproj_threshold,
                  int iterations, uint32_t seed, double* pose, uint8_t* inliers, int32_t* status, void* scratch);

/* ---------------------------------------------------------------------------------------------
 * Training side: backward of the fused graph ops + the loss head of train.py:307-320.
 * Gradients are fp32; `pq` is the forward's saved GEMM output in `dtype`.
 * ------------------------------------------------------------------------------------------- */

/* Backward of cp_edgeconv_gather_max == autograd of `x.max(dim=-1)` over get_graph_feature + LeakyReLU
 * (init.py:36-49,64-68): gq = gout * leaky'(max_k P' + Q');  dQ'[b,i] = gq;  dP'[b,j] = sum of gq over the edges
 * (i,k) with idx[i,k] == j that won the max (first arg-max).  gout (B,N,gout_cstride) fp32 (channels
 * [gout_coff, +C)), dpq (B,N,2C) fp32 fully overwritten.  rev_ptr (G,N+1) / rev_edge (G,N*K): the static graph's
 * reverse adjacency -- flat edge ids i*K+k stably sorted by idx[i,k] -- so the scatter is a deterministic gather.
 * workspace: cp_edgeconv_bwd_workspace_bytes(B,N,C) bytes. */
size_t cp_edgeconv_bwd_workspace_bytes(int B, int N, int C);
int cp_edgeconv_gather_max_bwd(cp_stream_t stream, int dtype, const void* pq, const int32_t* idx,
                               const int32_t* rev_ptr, const int32_t* rev_edge, const int32_t* graph_ids,
                               const float* gout, float* dpq, void* workspace, int B, int N, int K, int C,
                               int G, int gout_cstride, int gout_coff, float slope);

/* Backward of cp_index2feat_gather == autograd of the four advanced-index gathers + mask multiply
 * (pipeline.py:156-163,280): dpatches (B,Hp,Wp,E) fp32, zeroed then scatter-added (fp32 hardware atomics). */
int cp_index2feat_gather_bwd(cp_stream_t stream, const float* gout, const int32_t* x_id, const int32_t* y_id,
                             const float* mask, float* dpatches, int B, int N, int Hp, int Wp, int E, int k,
                             int gout_cstride, int gout_coff);
/* same with gout stored in `dtype` (the training program keeps activation gradients in the storage type) */
int cp_index2feat_gather_bwd_t(cp_stream_t stream, int dtype, const void* gout, const int32_t* x_id, const int32_t* y_id,
                               const float* mask, float* dpatches, int B, int N, int Hp, int Wp, int E, int k,
                               int gout_cstride, int gout_coff);

/* UnmaskedCodeLoss (mask == NULL; losses/code_loss.py:6-27) / MaskedCodeLoss (mask (B,N); code_loss.py:30-62):
 * pred (B,nbits,N) fp32 logits with batch stride pred_bstride (elements), gt likewise (train.py:312-313 passes the
 * slice pixel_x_codes[:, :num_proj_bits]).  loss_type CP_LOSS_BCE (BCEWithLogits) | CP_LOSS_L1 (L1 on sigmoid).
 * Writes loss[0] and, if dpred != NULL, d loss / d pred (same indexing, stride dpred_bstride).
 * workspace: cp_loss_workspace_bytes() bytes, 16-byte aligned. */
size_t cp_loss_workspace_bytes(void);
int cp_code_loss(cp_stream_t stream, int loss_type, const float* pred, long long pred_bstride, const float* gt,
                 long long gt_bstride, const float* mask, int B, int nbits, int N, float* loss, float* dpred,
                 long long dpred_bstride, void* workspace);

/* MaskedCodeLoss(loss_type="CE") (losses/code_loss.py:36-37,47-61; not used by train.py:87-88, kept for the class contract):
 * pred (B,C,N) fp32 class logits (batch stride pred_bstride), gt_class (B,N) class ids as fp32, mask (B,N):
 * loss = sum_bn mask * (logsumexp_c pred - pred[gt]) / clamp(sum mask, 1); dpred (optional) = d loss / d pred. */
int cp_masked_ce_loss(cp_stream_t stream, const float* pred, long long pred_bstride, const float* gt_class,
                      const float* mask, int B, int C, int N, float* loss, float* dpred, long long dpred_bstride,
                      void* workspace);

/* MaskLoss_interpolate (losses/mask_loss.py:6-17): mean | sigmoid(pred[b,0]) - nearest_resize(gt[b]) |.
 * pred points at channel 0 of the slice the caller passes (train.py:315-316: pred_seg[:, 0:1] / [:, 1:2]), (h,w)
 * contiguous with batch stride pred_bstride; gt (B,Hm,Wm) fp32 contiguous. */
int cp_mask_loss(cp_stream_t stream, const float* pred, long long pred_bstride, const float* gt, int B, int h,
                 int w, int Hm, int Wm, float* loss, float* dpred, long long dpred_bstride, void* workspace);

/* ---------------------------------------------------------------------------------------------
 * Training side, dense layers (reference train.py:319 `loss.backward()` through every
 * nn.Conv2d / nn.ConvTranspose2d / nn.Linear of the path).
 *
 * Weight gradient:  dw[dw_base + co*dw_sco + ci*dw_sci + r*dw_sr + s*dw_ss] +=
 *       sum_{b,oy,ox} dy[b,oy,ox,co] * x[b, oy*stride-pad+r, ox*stride-pad+s, ci]          (fp32 atomics)
 * dy (B,Ho,Wo,dy_cstride) and x (B,H,W,x_cstride) are channels-last `dtype` tensors (channel slices via *_coff);
 * the caller zeroes dw first.  nn.Conv2d weight (Cout,Cin,R,S): dw_sco = Cin*R*S, dw_sci = R*S, dw_sr = S, dw_ss = 1.
 * nn.ConvTranspose2d(k3,s2,p1,op1) weight (Cin_t,Cout_t,3,3): pass the layer INPUT as `dy` (coarse grid) and the
 * output gradient as `x` (fine grid), stride 2, pad 1 -- same strides.  Data-gradients need no entry point of their
 * own: they are cp_conv2d_igemm / cp_conv3x3_halo / cp_gemm_rows launches over dy with cp_weight_dgrad()'s weights.
 * ------------------------------------------------------------------------------------------- */
typedef struct CpWgradDesc {
  int32_t dtype;
  int32_t B, H, W;             /* x spatial size */
  int32_t Ho, Wo;              /* dy spatial size */
  int32_t Cout, dy_cstride, dy_coff;
  int32_t Cin, x_cstride, x_coff;
  int32_t R, S, stride, pad;
  int64_t dw_base, dw_sco, dw_sci, dw_sr, dw_ss;
} CpWgradDesc;
int cp_conv2d_wgrad(cp_stream_t stream, const CpWgradDesc* d, const void* dy, const void* x, float* dw);
/* same, with a caller-owned scratch buffer: the pixel-slice partial sums go through it (plain stores + one reduction
 * launch, deterministic) instead of through fp32 atomics -- measured ~30 G atomics/s made every launch cost >= 0.6 ms.
 * dw is still accumulated into (+=).  Any size works (it bounds the number of slices); 160 MiB never limits. */
int cp_conv2d_wgrad_ws(cp_stream_t stream, const CpWgradDesc* d, const void* dy, const void* x, float* dw, void* workspace,
                       size_t workspace_bytes);

/* Deferred reduction (training step: ~360 weight-gradient launches per backward, each followed by a latency-bound reduction of
 * its pixel-slice partials): cp_conv2d_wgrad_deferred launches ONLY the partial kernel into `workspace` (which must stay
 * untouched until the reduction ran: cp_conv2d_wgrad_scratch_bytes() bytes give the launcher its preferred slice count) and
 * fills *item on the HOST; cp_conv2d_wgrad_plan fills the same item without launching (static launch programs resolve it at
 * build time).  item->ws == NULL: the layer needs no reduction (atomics / direct stores).  cp_wgrad_reduce_batch runs any number
 * of owed reductions in ONE launch: items in DEVICE memory + the exclusive prefix sum (n + 1 entries) of
 * cp_wgrad_reduce_item_blocks() of each.  Sums in slice order: bit-identical to cp_conv2d_wgrad_ws. */
typedef struct CpWgradReduceItem {
  const float* ws; float* dw;
  int32_t S, GY, co_blocks, ci_blocks, R, Ssz, Cout, Cin, taps_in_block;
  long long dw_base, dw_sco, dw_sci, dw_sr, dw_ss;
} CpWgradReduceItem;
size_t cp_conv2d_wgrad_scratch_bytes(const CpWgradDesc* d);
int cp_conv2d_wgrad_deferred(cp_stream_t stream, const CpWgradDesc* d, const void* dy, const void* x, float* dw, void* workspace,
                             size_t workspace_bytes, CpWgradReduceItem* item);
int cp_conv2d_wgrad_plan(const CpWgradDesc* d, const void* dy, const void* x, float* dw, void* workspace, size_t workspace_bytes,
                         CpWgradReduceItem* item);
uint32_t cp_wgrad_reduce_item_blocks(const CpWgradReduceItem* item);
int cp_wgrad_reduce_batch(cp_stream_t stream, const CpWgradReduceItem* items_dev, const uint32_t* block_prefix_dev, int n_items,
                          uint32_t total_blocks);

/* Grouped weight gradients: the partial-sum launches of SEVERAL layers in one launch per kernel kind.  cp_conv2d_wgrad_item =
 * cp_conv2d_wgrad_plan that also describes the layer's partial-sum launch in `compute` (nothing is launched) and cuts the layer's
 * pixels for about `target_blocks` workgroups (0: a whole GPU's worth, as the single-layer launches do) -- a caller that groups
 * n layers hands each its share of ~512 workgroups, which shrinks the partial tiles by the same factor.  reduce->ws == NULL: the
 * layer adds with atomics, it owes no reduction.  cp_wgrad_group: `items_dev` = items of ONE kind in device memory, `prefix_dev`
 * the exclusive prefix sum (n_items + 1 entries) of item.blocks.  Results equal the single-layer launches' up to fp32 summation
 * order across slices. */
#define CP_WGRAD_ITEM_BYTES 160
enum { CP_WGRAD_ITEM_3X3 = 0, CP_WGRAD_ITEM_3X3_SMALL = 1, CP_WGRAD_ITEM_GENERIC_BF16 = 2, CP_WGRAD_ITEM_GENERIC_F32 = 3,
       CP_WGRAD_ITEM_3X3_S2_SMALL = 4 };      /* all-taps kernels: stride 1 (0, 1); stride 2, <= 32 x 32 channels (4) */
typedef struct CpWgradItem {
  int32_t kind;
  uint32_t blocks, gx, gy;
  unsigned long long params[CP_WGRAD_ITEM_BYTES / 8];      /* opaque: the kernel's parameter block */
} CpWgradItem;
int cp_conv2d_wgrad_item(const CpWgradDesc* d, const void* dy, const void* x, float* dw, void* workspace, size_t workspace_bytes,
                         int target_blocks, CpWgradItem* compute, CpWgradReduceItem* reduce);
int cp_wgrad_group(cp_stream_t stream, int kind, const CpWgradItem* items_dev, const uint32_t* prefix_dev, int n_items,
                   uint32_t total_blocks);

/* Optimizer step over ALL parameter tensors in one launch (the reference's train.py:244-246,320: optim.Adam(net.parameters(), lr) or
 * optim.SGD(..., momentum=0.9); optimizer.step() per batch).  items (device memory): parameter, its gradient and the optimizer state of
 * one tensor, fp32, n elements; prefix = exclusive prefix sum (n_items + 1) of cp_opt_item_blocks(n).  cp_adam_multi = torch.optim.Adam
 * (amsgrad off; weight_decay is L2, added to the gradient; `step` counts from 1 and must exceed every item's step0);
 * cp_sgd_multi = torch.optim.SGD (dampening 0, no nesterov; m = momentum buffer, first_step: buf := grad; momentum 0: m may be NULL). */
typedef struct CpOptItem {
  float* p; const float* g; float* m; float* v;
  uint64_t n;
  uint32_t step0, pad;       /* Adam: the tensor's own step count is `step` - step0 (torch counts per parameter) */
} CpOptItem;
uint32_t cp_opt_item_blocks(uint64_t numel);
int cp_adam_multi(cp_stream_t stream, const CpOptItem* items_dev, const uint32_t* prefix_dev, int n_items, uint32_t total_blocks, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step);
int cp_sgd_multi(cp_stream_t stream, const CpOptItem* items_dev, const uint32_t* prefix_dev, int n_items, uint32_t total_blocks, float lr,
                 float momentum, float weight_decay, int first_step);

/* Weights of the data-gradient of a stride-1 conv: w (Cout,Cin,R,S) fp32 -> wt (Cin,Cout,R,S) fp32 with both taps
 * flipped (wt[ci][co][r][s] = w[co][ci][R-1-r][S-1-s]); dx = conv(dy, wt, stride 1, pad R-1-pad).  (Stride-2 3x3
 * convs use the transposed=1 phase packing of cp_pack_conv_weight on w itself; ConvTranspose2d's data-gradient is a
 * plain stride-2 conv with w read as (Cout'=Cin_t, Cin'=Cout_t, 3, 3).) */
int cp_weight_dgrad(cp_stream_t stream, const float* w, int Cout, int Cin, int R, int S, float* wt);

/* Train-mode BatchNorm2d (nn.BatchNorm2d defaults: eps 1e-5, momentum 0.1; inside timm hrnet, init.py:60,
 * pipeline.py:51,189-208) over the M = B*H*W rows of a channels-last tensor, C logical channels.
 * cp_bn_train_stats: batch mean / biased variance (fp64 sums) -> scale = gamma*rstd, shift = beta - mean*scale (fp32
 * vectors of ceil16(C) entries, zero beyond C), mean / rstd saved for the backward; running_mean / running_var
 * updated in place with `momentum` (unbiased variance), NULL = skip.  gamma / beta NULL = 1 / 0.
 * workspace: cp_bn_workspace_bytes(C); one workspace may serve many layers on one stream. */
size_t cp_bn_workspace_bytes(int C);
int cp_bn_train_stats(cp_stream_t stream, int dtype, const void* x, int M, int C, int x_cstride, int x_coff,
                      const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                      float eps, float* scale, float* shift, float* mean, float* rstd, void* workspace);
/* y = act(x * scale[c] + shift[c] + res) elementwise (res optional; y may alias x or res). */
int cp_affine_act(cp_stream_t stream, int dtype, const void* x, int x_cstride, int x_coff, const float* scale,
                  const float* shift, const void* res, int res_cstride, int res_coff, void* y, int y_cstride, int y_coff,
                  int M, int C, int act, float slope);
/* Backward of y = act(BN(x) + res):  dz = dy * act'(y);  dres (+)= dz;  dgamma = sum dz*xhat;  dbeta = sum dz;
 * dx = gamma*rstd*(dz - mean(dz) - xhat*mean(dz*xhat)).  x == NULL selects the bias-only form (y = act(conv + b)):
 * dx = dz, dbeta = sum dz.  dx may alias dy.  workspace: cp_bn_bwd_workspace_bytes(C). */
size_t cp_bn_bwd_workspace_bytes(int C);
int cp_bn_train_bwd(cp_stream_t stream, int dtype, const void* dy, int dy_cstride, int dy_coff, const void* y,
                    int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff, const float* mean,
                    const float* rstd, const float* gamma, int M, int C, int act, float slope, void* dx, int dx_cstride,
                    int dx_coff, void* dres, int dres_cstride, int dres_coff, int dres_accumulate, float* dgamma,
                    float* dbeta, void* workspace);

/* Two-launch forms of the same BatchNorm forward / backward (what the training program uses): the column sums go into a
 * caller-zeroed fp64 accumulator block of cp_bn_acc_doubles(C) doubles (8 replicated [sum | sum of squares] resp.
 * [sum dz | sum dz*xhat] pairs of ceil16(C) entries: block b adds into replica b % 8 to spread the atomic traffic) by hardware
 * fp64 atomics, and the consumer launch derives the per-channel coefficients in its prologue -- no finalize launch (670 of them
 * were 8.5 % of a training step).  cp_bn_apply also writes mean / rstd (C floats each) and updates the running statistics;
 * cp_bn_bwd_apply writes dgamma / dbeta.  One accumulator pair per layer and pass; zero them all with one cp_memset_zero. */
size_t cp_bn_acc_doubles(int C);
int cp_bn_stats_accumulate(cp_stream_t stream, int dtype, const void* x, int M, int C, int x_cstride, int x_coff, double* acc);
int cp_bn_apply(cp_stream_t stream, int dtype, const void* x, int x_cstride, int x_coff, const double* acc, const float* gamma,
                const float* beta, float* running_mean, float* running_var, float momentum, float eps, const void* res,
                int res_cstride, int res_coff, void* y, int y_cstride, int y_coff, int M, int C, int act, float slope,
                float* mean, float* rstd);
int cp_bn_bwd_accumulate(cp_stream_t stream, int dtype, const void* dy, int dy_cstride, int dy_coff, const void* y,
                         int y_cstride, int y_coff, const void* x, int x_cstride, int x_coff, const float* mean,
                         const float* rstd, int M, int C, int act, float slope, double* acc);
int cp_bn_bwd_apply(cp_stream_t stream, int dtype, const void* dy, int dy_cstride, int dy_coff, const void* y, int y_cstride,
                    int y_coff, const void* x, int x_cstride, int x_coff, const float* mean, const float* rstd,
                    const float* gamma, const double* acc, int M, int C, int act, float slope, void* dx, int dx_cstride,
                    int dx_coff, void* dres, int dres_cstride, int dres_coff, int dres_accumulate, float* dgamma, float* dbeta);
/* The same two passes in ONE launch each (round 3): statistics over a block's rows -> grid barrier -> apply to the same rows.
 * `counter`: 4 zeroed bytes per call (zero them with the accumulators).  The grid (<= 512 blocks) is always co-resident on an
 * MI355X; the barrier's spin is bounded.  Arguments as cp_bn_stats_accumulate + cp_bn_apply / cp_bn_bwd_accumulate + cp_bn_bwd_apply. */
int cp_bn_train_fused(cp_stream_t stream, int dtype, const void* x, int x_cstride, int x_coff, double* acc, uint32_t* counter,
                      const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                      const void* res, int res_cstride, int res_coff, void* y, int y_cstride, int y_coff, int M, int C, int act,
                      float slope, float* mean, float* rstd);
int cp_bn_bwd_fused(cp_stream_t stream, int dtype, const void* dy, int dy_cstride, int dy_coff, const void* y, int y_cstride,
                    int y_coff, const void* x, int x_cstride, int x_coff, const float* mean, const float* rstd, const float* gamma,
                    double* acc, uint32_t* counter, int M, int C, int act, float slope, void* dx, int dx_cstride, int dx_coff,
                    void* dres, int dres_cstride, int dres_coff, int dres_accumulate, float* dgamma, float* dbeta);

/* Grouped BatchNorm passes: the same pass (statistics / apply / backward sums / backward apply) of up to CP_BN_GROUP_MAX
 * INDEPENDENT layers in ONE launch -- the branches of an HRNet module at equal depth (timm HighResolutionModule.forward runs
 * them one after the other; at the reference's training batch of 32 each pass is a 5-13 us launch over 0.3-5 MB).
 * cp_bn_item_* take the arguments of cp_bn_stats_accumulate / cp_bn_apply / cp_bn_bwd_accumulate / cp_bn_bwd_apply, run the
 * same checks and fill one item on the HOST (nothing is launched).  cp_bn_group: `items_dev` = the items of ONE kind and
 * dtype copied to device memory, `prefix_dev` = exclusive prefix sum (n_items + 1 entries) of item.blocks, `lds_bytes` = the
 * largest item.lds_bytes.  Every item behaves exactly as its single-layer launch (same block plan, same arithmetic). */
#define CP_BN_ITEM_BYTES 192
#define CP_BN_GROUP_MAX 16
enum { CP_BN_ITEM_STATS = 0, CP_BN_ITEM_APPLY = 1, CP_BN_ITEM_BWD_SUMS = 2, CP_BN_ITEM_BWD_APPLY = 3 };
typedef struct CpBnItem {
  int32_t kind, dtype;
  uint32_t blocks, lds_bytes;
  unsigned long long params[CP_BN_ITEM_BYTES / 8];      /* opaque: the kernel's parameter block */
} CpBnItem;
int cp_bn_item_stats(int dtype, const void* x, int M, int C, int x_cstride, int x_coff, double* acc, CpBnItem* item);
int cp_bn_item_apply(int dtype, const void* x, int x_cstride, int x_coff, const double* acc, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, const void* res, int res_cstride,
                     int res_coff, void* y, int y_cstride, int y_coff, int M, int C, int act, float slope, float* mean, float* rstd,
                     CpBnItem* item);
int cp_bn_item_bwd_sums(int dtype, const void* dy, int dy_cstride, int dy_coff, const void* y, int y_cstride, int y_coff,
                        const void* x, int x_cstride, int x_coff, const float* mean, const float* rstd, int M, int C, int act,
                        float slope, double* acc, CpBnItem* item);
int cp_bn_item_bwd_apply(int dtype, const void* dy, int dy_cstride, int dy_coff, const void* y, int y_cstride, int y_coff,
                         const void* x, int x_cstride, int x_coff, const float* mean, const float* rstd, const float* gamma,
                         const double* acc, int M, int C, int act, float slope, void* dx, int dx_cstride, int dx_coff, void* dres,
                         int dres_cstride, int dres_coff, int dres_accumulate, float* dgamma, float* dbeta, CpBnItem* item);
int cp_bn_group(cp_stream_t stream, int dtype, int kind, const CpBnItem* items_dev, const uint32_t* prefix_dev, int n_items,
                uint32_t total_blocks, uint32_t lds_bytes);

/* Train-mode EdgeConv in factored form (StaticGraph_module init.py:54-68 with BatchNorm2d batch statistics over the
 * B*N*K edges), see csrc/train_edge.hip.  pq (B,N,2C) = raw node GEMM output [P | Q] (W rows [W1 ; W2-W1], no
 * affine); kstar (B,N,C) uint8 receives the arg-max neighbour slot; scale/shift/mean/rstd: C floats each.
 * Backward writes dpq (B,N,2C) `dtype` = [dP - dQ | dQ], the operand of the node GEMM's weight- and data-gradient
 * (cp_edge_weight_view mode 1), plus dgamma / dbeta.  rev_ptr (G,N+1), rev_edge (G,N*K): reverse adjacency as for
 * cp_edgeconv_gather_max_bwd.  workspace: cp_edge_train_workspace_bytes(B, C).  C % 16 == 0. */
size_t cp_edge_train_workspace_bytes(int B, int C);
int cp_edgeconv_train_fwd(cp_stream_t stream, int dtype, const void* pq, const int32_t* idx, const int32_t* graph_ids,
                          const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                          float eps, void* out, int out_cstride, int out_coff, uint8_t* kstar, float* scale, float* shift,
                          float* mean, float* rstd, void* workspace, int B, int N, int K, int C, int G, float slope);
int cp_edgeconv_train_bwd(cp_stream_t stream, int dtype, const void* pq, const int32_t* idx, const int32_t* rev_ptr,
                          const int32_t* rev_edge, const int32_t* graph_ids, const void* out, int out_cstride, int out_coff,
                          const uint8_t* kstar, const void* gout, int gout_cstride, int gout_coff, const float* gamma,
                          const float* mean, const float* rstd, void* dpq, float* dgamma, float* dbeta, void* workspace,
                          int B, int N, int K, int C, int G, float slope);
/* EdgeConv weight views: w (C', 2C) = [W1 | W2] -> mode 0: (2C', C) = [W1 ; W2 - W1] (forward node GEMM);
 * mode 1: (C, 2C') with out[c][c'] = W1[c'][c], out[c][C'+c'] = W2[c'][c] (data-gradient of the node GEMM over D). */
int cp_edge_weight_view(cp_stream_t stream, const float* w, int Cout, int Cin, int mode, float* out);

/* Backward of cp_upsample2x_bilinear_ac (gather form, deterministic) and of ONE source of cp_fuse_sum_act
 * (dsrc (+)= sum over the 2^shift x 2^shift block of dout * [out > 0]). */
int cp_upsample2x_bilinear_ac_bwd(cp_stream_t stream, int dtype, const void* dout, void* din, int B, int H, int W, int C,
                                  int out_cstride, int out_coff, int in_cstride, int in_coff, int accumulate);
int cp_fuse_sum_act_bwd(cp_stream_t stream, int dtype, const void* dout, const void* out, void* dsrc, int B, int Hs, int Ws,
                        int C, int shift, int relu, int accumulate);
/* ... and every (output, term) pair of one fuse layer in ONE launch (the pairs write distinct gradient tensors): the item is
 * cp_fuse_sum_act_bwd's argument list checked and packed on the host, `blocks` its workgroup count; items_dev / prefix_dev as for
 * the other grouped launches (<= 64 items). */
typedef struct CpFuseBwdItem {
  const void* dout; const void* out; void* dsrc;
  int32_t Hs, Ws, CG, shift, relu, accumulate;
  uint64_t total;
} CpFuseBwdItem;
int cp_fuse_sum_act_bwd_item(int dtype, const void* dout, const void* out, void* dsrc, int B, int Hs, int Ws, int C, int shift, int relu,
                             int accumulate, CpFuseBwdItem* item, uint32_t* blocks);
int cp_fuse_sum_act_bwd_group(cp_stream_t stream, int dtype, const CpFuseBwdItem* items_dev, const uint32_t* prefix_dev, int n_items,
                              uint32_t total_blocks);
/* Backward of cp_maxpool3x3s2 (resnet34 stem): x (B,H,W,C) forward input, dout (B,H/2,W/2,C), din (+)= routed gradient
 * (first maximum of each window in row-major order takes it, as ATen's CPU kernel does). */
int cp_maxpool3x3s2_bwd(cp_stream_t stream, int dtype, const void* x, const void* dout, void* din, int B, int H, int W, int C,
                        int accumulate);

/* plumbing: stream-ordered zero fill / device copy (capturable), and a tensor (fp32 or `dtype`: src_dtype) with arbitrary element strides (the logit block,
 * NCHW seg logits, fp32 scatter targets) -> channels-last `dtype` (B, HW, Cphys) with zero padded channels. */
int cp_memset_zero(cp_stream_t stream, void* p, size_t nbytes);
int cp_strided_to_nhwc(cp_stream_t stream, int dtype, const void* src, int src_dtype, long long base, long long sb,
                       long long sp, long long sc, void* out, int B, int HW, int C, int Cphys);
int cp_memcpy_d2d(cp_stream_t stream, void* dst, const void* src, size_t nbytes);

/* ---------------------------------------------------------------------------------------------
 * Batched weight preparation (training): a step re-packs ~1100 live weight tensors; as individual ~4 us launches that
 * was a fifth of the step.  cp_pack_item_*() fill one CpPackItem each on the HOST (same arguments and the same packed
 * image as the single-launch entry points they mirror), the caller keeps the items in DEVICE memory together with an
 * exclusive prefix sum of their 256-thread block counts, and cp_pack_batch() processes all of them in ONE launch.
 * Items of one batch must be independent: views (dgrad / EdgeConv / copies) that feed packs go into an earlier batch.
 * ------------------------------------------------------------------------------------------- */
enum { CP_PACK_GENERIC = 0, CP_PACK_HALO_S = 1, CP_PACK_HALO = 2, CP_PACK_HALO4 = 3, CP_PACK_GEMM = 4,
       CP_PACK_DGRAD_VIEW = 5, CP_PACK_EDGE_VIEW = 6, CP_PACK_COPY_F32 = 7 };
typedef struct CpPackItem {
  int32_t kind;
  int32_t a[9];
  const void* src;
  void* dst;
  const int32_t* row_map;
  uint64_t total;              /* elements = threads of this item */
} CpPackItem;
int cp_pack_item_conv(int dtype, const float* w, int Cout, int Cin, int R, int S, int cin_phys, int transposed, int phase,
                      const int32_t* row_map, int cout_rows, void* packed, CpPackItem* item);      /* cp_pack_conv_weight */
int cp_pack_item_halo(int dtype, const float* w, int Cout, int Cin, int cin_phys, void* packed, CpPackItem* item);  /* cp_pack_conv3x3_halo_weight */
int cp_pack_item_gemm(int dtype, const float* w, int Cout, int Cin, int cin_phys, void* packed, CpPackItem* item);  /* cp_pack_gemm_weight */
int cp_pack_item_dgrad_view(const float* w, int Cout, int Cin, int R, int S, float* wt, CpPackItem* item);          /* cp_weight_dgrad */
int cp_pack_item_edge_view(const float* w, int Cout, int Cin, int mode, float* out, CpPackItem* item);              /* cp_edge_weight_view */
int cp_pack_item_copy_f32(const float* src, float* dst, int count, CpPackItem* item);
int cp_pack_batch(cp_stream_t stream, int dtype, const CpPackItem* items_dev, const uint32_t* block_prefix_dev, int n_items,
                  uint32_t total_blocks);

/* layout plumbing at the boundary: NCHW fp32 image -> channels-last `dtype` (C padded with zeros to
 * Cphys), and channels-last slice -> NCHW fp32 (for `return_img_feats`, init.py:123-124). */
int cp_nchw_to_nhwc(cp_stream_t stream, int dtype, const float* in, void* out, int B, int C, int H, int W,
                    int Cphys);
int cp_nhwc_to_nchw_f32(cp_stream_t stream, int dtype, const void* in, float* out, int B, int C, int H, int W,
                        int in_cstride, int in_coff);

/* Input side on the device (next-row N3; reference bop_dataset_pytorch.py:385-391 ToTensor + Normalize): uint8
 * (B,H,W,3) crop -> (x/255 - mean)/std -> channels-last `dtype` with Cphys channels (zero padded).  mean3/std3 are
 * HOST pointers to 3 floats. */
int cp_u8hwc_to_nhwc_norm(cp_stream_t stream, int dtype, const uint8_t* in, void* out, int B, int H, int W, int Cphys,
                          const float* mean3, const float* std3);

/* Input side, second half of row N3: the data loader's RoI crops (bop_dataset_pytorch.py:132-145 get_roi with resize_method
 * crop_square_resize :55-91 or crop_resize :94-108, i.e. zero-padded window + cv2.resize) for a whole batch from full uint8 images
 * (n_img, H, W, C <= 4) resident in device memory.  windows (device, B x 6 int32): x1, y1, x2, y2, roi_w, roi_h -- roi pixel
 * (ry, rx) = image pixel (y1 + ry, x1 + rx) inside [max(x1,0), min(x2,W)) x [max(y1,0), min(y2,H)), zero elsewhere; the roi is resized
 * to crop x crop with cv2's 8-bit INTER_NEAREST (0) / INTER_LINEAR (1) arithmetic.  img_idx (device, B) or NULL when n_img is 1 or B.
 * out: (B, crop, crop, C) uint8, the operand of cp_u8hwc_to_nhwc_norm.  An empty roi gives a zero crop. */
int cp_crop_resize_u8(cp_stream_t stream, const uint8_t* images, int n_img, int H, int W, int C, const int32_t* windows,
                      const int32_t* img_idx, uint8_t* out, int B, int crop, int interpolation);

/* ---------------------------------------------------------------------------------------------
 * hipGraph helpers: capture the launch sequence of one forward (everything above is capture-safe:
 * no allocation, no synchronisation) and replay it with one call.
 * ------------------------------------------------------------------------------------------- */
int cp_graph_begin_capture(cp_stream_t stream);
int cp_graph_end_capture(cp_stream_t stream, void** graph_exec_out);
/* Dataflow capture on ONE stream: name the graph nodes the next launch depends on (n = 0: none, a root of the graph), and
 * read back the node(s) the stream's next launch would wait for (after a launch: that launch's last kernel node). */
int cp_graph_capture_set_deps(cp_stream_t stream, void* const* nodes, int n);
int cp_graph_capture_tail(cp_stream_t stream, void** nodes_out, int cap, int* n_out);
int cp_graph_launch(void* graph_exec, cp_stream_t stream);
int cp_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* CHECKERPOSE_HIP_H */
