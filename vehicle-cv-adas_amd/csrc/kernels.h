// kernels.h -- launch interfaces of the network kernels (conv_kernels.hip, aux_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace adas {

enum { PREC_BF16 = 0, PREC_FP32 = 1, PREC_FP16 = 2, PREC_X3 = 3 };  // == ADAS_PREC_* (include/adas_hip.h)
inline bool prec_is16(int prec) { return prec == PREC_BF16 || prec == PREC_FP16; }   // bf16 and fp16 share every 16-bit kernel (elem16.h)
// bytes per activation / weight element: the split precision stores a (hi, lo) pair of halves per element (elem16.h, x3s)
inline int prec_esize(int prec) { return prec_is16(prec) ? 2 : 4; }
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_LEAKY = 3 /* LeakyReLU(0.1): YOLOv7 (conv_halo, conv_pw, conv_pwg, conv_igemm) */,
       // element-wise only (OP_WSUM with one input = a stand-alone activation layer; OP_SE_GATE's gate): the conv epilogues do not carry them and
       // the engine refuses a convolution that asks for one (PP-LCNet / MobileNetV3-style networks: torch.nn.Hardswish / Hardsigmoid)
       ACT_HSWISH = 4 /* x relu6(x + 3) / 6 */, ACT_HSIGMOID = 5 /* relu6(x + 3) / 6 */, ACT_RELU6 = 6 /* min(max(x, 0), 6): MobileNetV2-style backbones */ };
enum { RES_NONE = 0, RES_AFTER_ACT = 1, RES_BEFORE_ACT = 2 };

// Workgroups per XCD the PERSISTENT kernels (conv_halo8, conv_halo_rw, conv_stem: one resident workgroup set walking a work list) launch:
// 32 = every CU (default).  ADAS_PERSIST_SLOTS=<n> leaves (32 - n) CUs of every XCD to whatever else is in flight (the other
// network's kernels on the pipeline's second stream) -- an experiment knob, read once.
int persist_slots(int which);   // 0 conv_halo8, 1 conv_halo_rw, 2 conv_stem

// One NHWC tensor view: channel slice [coff, coff+c) of a buffer whose pixel stride is `cs` elements.
struct TView {
    void* p;
    int cs, coff, c;
    int h, w;
    int f32;  // storage is fp32 regardless of the engine precision
};

struct ConvArgs {
    TView in, out, res;
    const void* wgt;    // [cout_pad][kpad] in the compute type, K = (r*kw+s)*cin + c
    const float* bias;  // [cout_pad]
    int n;              // batch
    int kh, kw, stride, pad, act, res_mode;
    int k, kpad, m;     // K = kh*kw*cin ; padded to 32 ; M = n*ho*wo
    int max_n;          // the engine's max_batch (static: decides the FC weight packing)
    int prec;           // PREC_*: the 16-bit kernels pick their operand type (bf16 | fp16) from it
    // 1x1 convs over a concat whose FIRST up_c channels are a 2x nearest-neighbour upsample of `up`: conv_pw.hip reads those
    // channels from the half-resolution tensor at (y / 2, x / 2) and the upsample launch is dropped (up_c = 0: none)
    TView up{};
    int up_c = 0;
    // conv_halo8.hip: the residual of this 3x3 conv is a 1x1 stride-2 projection of ds_in (ResNet layerN.0 shortcut); when ds_w is set
    // the projection is accumulated inside this launch (weights [cout / 64][ds_in.c / 32][64][32], bias ds_bias) and `res` is ignored
    TView ds_in{};
    const void* ds_w = nullptr;
    const float* ds_bias = nullptr;
    int halo_bn = 0;    // conv_halo: output channels per workgroup the weights were packed for (0: halo_bn(cout); see plan_halo_bn)
    // split precision: the same weights in conv_halo8_x3.hip's half-chunk slab packing (nullptr: the layer's shape does not take it)
    const void* wgt_h8x3 = nullptr;
};

// Which kernel runs a conv and how its weights are packed.  Decided once at load time from static
// shapes and re-derived identically at launch time.
enum { CONV_GATHER = 0, CONV_HALO = 1, CONV_FC = 2, CONV_STEM = 3, CONV_PW = 4, CONV_STEM2 = 5 /* second conv of a fused YOLO stem */,
       CONV_PAIR = 6 /* either conv of a fused 3x3 -> 3x3 pair (conv_pair.hip) */,
       CONV_C2F_PW = 7 /* the 1x1 convs of a fused C2f block (conv_c2f.hip): MFMA-fragment packing */,
       CONV_DET5 = 8 /* the per-level 1x1 of a v5-layout Detect folded into the decode launch (aux_kernels.hip): per-anchor fragments */ };
struct ConvPlan {
    int kernel;   // CONV_*
    int cin_pad;  // channels per tap in the packed weights (halo: padded to 32 so the tail is zero)
    int kpad;     // packed K extent = round32(kh*kw*cin_pad)
};
ConvPlan plan_conv(int prec, int kh, int kw, int stride, int pad, int max_n, int res_mode, const TView& in, const TView& out);

// conv_halo_rw.hip: persistent, weights-resident variant of the stride-1 halo kernel for Cin <= 64 (same weight packing)
bool halo_rw_applicable(int kh, int kw, int stride, int pad, int n, const TView& in, const TView& out);
hipError_t launch_conv_halo_rw(const ConvArgs& a, hipStream_t st);
// conv_halo_s2.hip: stride-2 3x3 for Cout % 128 == 0 (parity-plane LDS window, 8 waves, same weight packing)
bool halo_s2p_applicable(int kh, int kw, int stride, int pad, int res_mode, int n, const TView& in, const TView& out);
hipError_t launch_conv_halo_s2p(const ConvArgs& a, hipStream_t st);
// ... and its split-precision form on conv_halo8_x3's weight slabs (ConvArgs::wgt_h8x3): shape_ok decides the packing at load time
bool halo_s2p_x3_shape_ok(int kh, int kw, int stride, int pad, int res_mode, const TView& in, const TView& out);
bool halo_s2p_x3_applicable(int kh, int kw, int stride, int pad, int res_mode, int n, const TView& in, const TView& out);
hipError_t launch_conv_s2p_x3(const ConvArgs& a, hipStream_t st);   // (takes the layer to conv_s2d_x3_kernel, the LDS-DMA form, where that applies)
bool halo_s2d_x3_applicable(int kh, int kw, int stride, int pad, int res_mode, int n, const TView& in, const TView& out);
// conv_halo.hip's tile plan (strip-linear tiles of halo_bm(S) output pixels), shared with conv_halo8.hip
struct HaloPlan {
    int SW, NS, TPS, WW, maxpix;   // strip width, strips per row, tiles per strip, window width, window pixels
    double eff;                    // useful fraction of the tiles' pixels
    uint32_t mg_ww, mg_sw;         // n / WW == (n * mg_ww) >> 20, n / SW == (n * mg_sw) >> 20 for every n the kernels divide
};
// maxpix_cap: window pixel budget (0: the kernel's default); bm: output pixels per tile (0: halo_bm(S) = 256 at stride 1, 128 at stride 2)
// policy: 0 = most efficient strip width, the widest on ties; 1 = the smallest window among the widths within 0.5 % of the best efficiency
bool plan_halo(int Ho, int Wo, int S, HaloPlan* out, int maxpix_cap = 0, int bm = 0, int policy = 0);
bool plan_halo_sw(int Ho, int Wo, int S, int SW, int maxpix_cap, HaloPlan* out, int bm = 0);   // one given strip width
int halo_tile_pixels(const ConvArgs& a);   // conv_halo.hip: the tile size launch_conv_halo picks for this launch (128 on small stride-1 layers)
// conv_halo8.hip: stride-1 3x3 for Cout % 128 == 0, Cin % 32 == 0: persistent, LDS-DMA fed, counted waits (same weight packing)
bool halo8_applicable(int kh, int kw, int stride, int pad, int n, const TView& in, const TView& out, const TView& res, int res_mode);
// the same conv with its projection shortcut (1x1 stride 2 on `x`, no activation) folded in
bool halo8_ds_applicable(int kh, int kw, int stride, int pad, int n, const TView& in, const TView& out, const TView& x);
hipError_t launch_pack_weights_ds(const float* src, void* dst, int cout, int cin, int prec, hipStream_t st);   // src fp32 [cout][cin]
hipError_t launch_conv_halo8(const ConvArgs& a, hipStream_t st);
// conv_pair.hip: conv A (x -> t) and conv B (t -> y [+ x]) in one launch, t never written: 3x3 s1 SiLU on 16 or 32 channels
bool pair_applicable(int prec, int kh, int kw, int stride, int pad, int act, int res_mode, const TView& x, const TView& t, int kh2, int kw2,
                     int stride2, int pad2, int act2, int res_mode2, const TView& y, const TView& res2);
hipError_t launch_pack_weights_pair(const float* src, void* dst, int c, int prec, hipStream_t st);   // src fp32 [c][9][c]
hipError_t launch_conv_pair(const TView& x, const TView& y, const void* w1, const float* b1, const void* w2, const float* b2, int n, bool has_res,
                            int prec, hipStream_t st);
// conv_c2f.hip: a whole C2f(32, 32, n = 1, shortcut) block (YOLOv8n / YOLOv10n model.2) in one launch: cv1 1x1 -> split -> 3x3 pair + shortcut ->
// cv2 1x1 over the 48-channel concat, which is never written.  cat01 / y1 / y2 / cat: the concat buffer's slices the separate ops use.
bool c2f16_applicable(int prec, const TView& x, const TView& cat01, const TView& y1, const TView& y2, const TView& cat, const TView& out);
hipError_t launch_pack_weights_c2f_pw(const float* src, void* dst, int cout, int cin, int prec, hipStream_t st);   // 1x1 weights as MFMA fragments
size_t c2f_pw_weight_bytes(int cout, int cin);
// conv_c2f_x3.hip: the same block in the split precision (hi / lo planes in LDS, three MFMAs per product, x3_silu); its Bottleneck pair is
// recognised by the pair pass (pair_x3_candidate) and released again if the C2f pass does not absorb it -- no stand-alone x3 pair kernel exists
bool pair_x3_candidate(int prec, int kh, int kw, int stride, int pad, int act, int res_mode, const TView& x, const TView& t, int kh2, int kw2, int stride2,
                       int pad2, int act2, int res_mode2, const TView& y);
bool c2f16_x3_applicable(int prec, const TView& x, const TView& cat01, const TView& y1, const TView& y2, const TView& cat, const TView& out);
hipError_t launch_pack_weights_pair16_x3(const float* src, void* dst, hipStream_t st);                       // src fp32 [16][9][16] -> [hi | lo] fragments
hipError_t launch_pack_weights_c2f_pw_x3(const float* src, void* dst, int cout, int cin, hipStream_t st);   // 1x1 weights -> [hi | lo] fragments
hipError_t launch_conv_c2f16_x3(const TView& x, const TView& out, const void* const w[4], const float* const b[4], int n, hipStream_t st);   // cv1, A, B, cv2
hipError_t launch_conv_c2f16(const TView& x, const TView& out, const void* w_cv1, const float* b_cv1, const void* w_a, const float* b_a, const void* w_b,
                             const float* b_b, const void* w_cv2, const float* b_cv2, int n, int prec, hipStream_t st);
bool pw_applicable(int prec, int kh, int kw, int stride, int pad, int res_mode, const TView& in, const TView& out);  // conv_pw.hip
// conv_pwg.hip: 1x1 stride-1 convs conv_pw does not take (Cin > 512), a K-looped MFMA GEMM on the generic [cout_pad][K] weight packing
bool pwg_applicable(int prec, int kh, int kw, int stride, int pad, const TView& in, const TView& out, const TView& res, int res_mode);
hipError_t launch_conv_pwg(const ConvArgs& a, hipStream_t st);
const char* pwg_kernel_name(int m, int cout);
// conv_x3.hip: the split precision's convolution (any kernel size / stride, channel counts multiples of 8) and its G8 weight packing
hipError_t launch_conv_x3(const ConvArgs& a, hipStream_t st);
const char* conv_x3_kernel_name(const ConvArgs& a);
hipError_t launch_pack_weights_x3(const float* src, void* dst, int cout, int cout_pad, int taps, int cin, int cin_pad, int kpad, hipStream_t st);
// conv_halo8_x3.hip: stride-1 3x3, Cout % 64 == 0, Cin % 32 == 0, in the split precision: persistent, LDS-DMA fed, half-chunk stream
bool halo8_x3_shape_ok(int kh, int kw, int stride, int pad, const TView& in, const TView& out);   // static: gets the second weight packing
bool halo8_x3_applicable(int kh, int kw, int stride, int pad, int n, const TView& in, const TView& out, const TView& res, int res_mode);
size_t halo8_x3_weight_bytes(int cout, int cin);
hipError_t launch_pack_weights_h8x3(const float* src, void* dst, int cout, int cin, hipStream_t st);   // src fp32 [cout][9][cin]
hipError_t launch_conv_halo8_x3(const ConvArgs& a, hipStream_t st);
// returns hipSuccess or the launch error.  prec: PREC_*.
hipError_t launch_conv(const ConvArgs& a, int prec, hipStream_t st);
const char* conv_tile_name(const ConvArgs& a, int prec);
const char* conv_kernel_name(const ConvArgs& a, int prec, int kernel);

hipError_t launch_input_nchw(const float* nchw, TView out, int n, int c_true, int prec, hipStream_t st);
hipError_t launch_maxpool(TView in, TView out, int n, int k, int s, int p, int prec, hipStream_t st);
hipError_t launch_avgpool(TView in, TView out, int n, int k, int s, int p, int prec, hipStream_t st);   // count_include_pad average (YOLOv9 AConv)
hipError_t launch_upsample2(TView in, TView out, int n, int prec, hipStream_t st);
bool depth2space_supported(const TView& in, const TView& out);
hipError_t launch_depth2space(TView in, TView out, int n, int prec, hipStream_t st);   // block 2: ConvTranspose2d(k 2, s 2) behind a 1x1 conv (YOLOv6)
hipError_t launch_detect_v6(const TView* ins, float* out, int n, int nc, int A, const int strides[3], hipStream_t st);
// three chained 5x5 s1 p2 max-pools (SPPF) in one launch: out[0] = pool(in), out[1] = pool(out[0]), out[2] = pool(out[1])
bool sppf_pool3_applicable(int prec, const TView& in, const TView out[3]);
hipError_t launch_sppf_pool3(const TView& in, const TView out[3], int n, int prec, hipStream_t st);
// YOLOv8 Detect decode: ins = {box0, cls0, box1, cls1, box2, cls2} fp32 logits NHWC; out fp32 [n][4+nc][A]
hipError_t launch_detect_v8(const TView* ins, float* out, int n, int nc, int A, const int strides[3], hipStream_t st);
hipError_t launch_detect_v8_fused(const TView* hidden, const void* const* wfrag, const float* const* bias, float* out, int n, int nc, int A,
                                  const int strides[3], int prec, hipStream_t st, float* sink_conf = nullptr, int* sink_cls = nullptr);
// the same fusion in the split precision (conv_pw_x3's weight packing as it is; exact DFL / sigmoid); kt_box / kt_cls: the two packings' K steps
hipError_t launch_detect_v8_fused_x3(const TView* hidden, const void* const* wfrag, const float* const* bias, int kt_box, int kt_cls, float* out, int n, int nc,
                                     int A, const int strides[3], hipStream_t st_, float* sink_conf = nullptr, int* sink_cls = nullptr);
// YOLOv5 Detect decode: ins = 3 fp32 maps [n][ny][nx][3*(5+nc)]; out fp32 [n][A][5+nc]; anchors[18] device
bool det5_applicable(int prec, int nc, const TView& in, const TView& logits);
size_t det5_weight_bytes(int no, int cin);
hipError_t launch_pack_weights_det5(const float* src, void* dst, int no, int cin, int prec, hipStream_t st);   // src fp32 [3 * no][cin]
hipError_t launch_detect_v5_fused(const TView* hidden, const void* const* wfrag, const float* const* bias, float* out, int n, int nc, int A,
                                  const int strides[3], const float* d_anchors, int prec, hipStream_t st, float* sink_conf = nullptr, int* sink_cls = nullptr);
hipError_t launch_detect_v5(const TView* ins, float* out, int n, int nc, int A, const int strides[3],
                            const float* d_anchors, hipStream_t st);
// LayerNorm over the flat per-frame vector (len elements, fp32 in) -> compute type out
hipError_t launch_layernorm(const float* in, void* out, const float* gamma, const float* beta, int n, int len, float eps,
                            int prec, hipStream_t st);
// fp32 -> compute-type weight packing on the device: src [cout][taps][cin] fp32,
// dst [cout_pad][kpad] with element (row, tap*cin_pad + c), zero padded
hipError_t launch_pack_weights(const float* src, void* dst, int cout, int cout_pad, int taps, int cin, int cin_pad, int kpad,
                               int prec, hipStream_t st);
// Fused first layer (conv_stem.hip): NCHW fp32 seam tensor -> stride-2 conv + act [+ 3x3 s2 p1 max-pool] -> NHWC bf16.
bool stem_applicable(int prec, int in_c_true, int kh, int kw, int stride, int pad, int act, int res_mode, const TView& out, bool pool,
                     const TView& pool_out);
size_t stem_weight_bytes(int kh, int cout);
// YOLO stem + the 3x3 s2 16->32 conv behind it in one launch (conv_stem.hip, CONV2)
size_t stem2_weight_bytes();
void stem2_pack_weights(const float* w_ohwi_32x3x3x16, uint16_t* dst_host, int prec);
bool stem2_applicable(int prec, int kh, int pad, int act, const TView& stem_out, int kh2, int kw2, int stride2, int pad2, int act2, int res_mode2,
                      const TView& out2);
hipError_t launch_conv_stem2(const float* nchw, int n, int c_true, int H, int W, int kh, int pad, const void* wfrag, const float* bias,
                             const TView& stem_out, const void* wfrag2, const float* bias2, const TView& out2, bool packed_in, int prec, hipStream_t st);
void stem_pack_weights(const float* w_ohwi, int cout, int kh, int kw, int cs, int c_true, uint16_t* dst_host, int prec);
// conv_stem_x3.hip: the first layer in the split precision (NCHW fp32 -> stride-2 conv + act -> G8 NHWC; no pool / second-conv fusion)
bool stem_x3_applicable(int in_c_true, int kh, int kw, int stride, int pad, int act, int res_mode, const TView& out);
size_t stem_x3_weight_bytes(int kh, int cout);
void stem_x3_pack_weights(const float* w_ohwi, int cout, int kh, int kw, int cs, int c_true, uint16_t* dst_host);   // hi array, then lo array
hipError_t launch_conv_stem_x3(const float* nchw, int n, int c_true, int H, int W, int kh, int pad, int act, const void* wfrag, const float* bias,
                               const TView& out, hipStream_t st);
// the ResNet stem (7x7 s2 + ReLU, 64 channels) with its 3x3 s2 p1 max-pool in one launch; same weight packing
// ... and the YOLO stems with the 3x3 s2 16 -> 32 conv behind them (model.1) in one launch (conv_stem2_x3_kernel)
bool stem2_x3_applicable(int in_c_true, int kh, int pad, int act, const TView& stem_out, int kh2, int kw2, int stride2, int pad2, int act2, int res_mode2,
                         const TView& out2);
size_t stem2_x3_weight_bytes();
void stem2_x3_pack_weights(const float* w_ohwi_32x3x3x16, uint16_t* dst_host);   // [hi | lo] fragment arrays
hipError_t launch_conv_stem2_x3(const float* nchw, int n, int c_true, int H, int W, int kh, int pad, const void* wfrag, const float* bias,
                                const TView& stem_out, const void* wfrag2, const float* bias2, const TView& out2, hipStream_t st);
bool stem_pool_x3_applicable(int in_c_true, int kh, int kw, int stride, int pad, int act, int res_mode, const TView& conv_out, const TView& pool_out);
hipError_t launch_conv_stem_pool_x3(const float* nchw, int n, int c_true, int H, int W, int pad, const void* wfrag, const float* bias,
                                    const TView& conv_out, const TView& pool_out, hipStream_t st);
hipError_t launch_conv_stem(const float* nchw, int n, int c_true, int H, int W, int kh, int pad, int act, const void* wfrag,
                            const float* bias, const TView& conv_out, bool pool, const TView& pool_out, bool packed_in, int prec, hipStream_t st);
// CONV_HALO packing: slab order [cout tile of halo_bn(cout)][32-channel chunk][tap][n within tile][32] -- the 9*BN*64 B a
// workgroup stages per chunk are one contiguous run (every wave-level staging load reads 1 KB of consecutive bytes).
// 65..96 output channels (YOLOv8n's class branch: 80) run as two 48-wide blocks: as two 64-wide blocks the second is mostly padding
// (an 80-wide single block was tried: 80 accumulators + 12 weight staging slots spill 54-94 VGPRs at two workgroups per CU).
inline int halo_bn(int cout) { return cout <= 16 ? 16 : (cout <= 32 ? 32 : ((cout > 64 && cout <= 96) ? 48 : 64)); }
// Output channels per conv_halo workgroup for a layer whose tiles do not fill the chip at the engine's max_batch (one frame at a time:
// 40x40x256 -> 256 is 13 tiles x 4 blocks = 52 workgroups, each an 8-chunk latency chain): narrower blocks = more, shorter workgroups.
// 0 = halo_bn(cout).  Decided at load (it fixes the weight packing); a layer with a non-default block runs on conv_halo only.
int plan_halo_bn(int max_n, int stride, const TView& in, const TView& out);
hipError_t launch_pack_weights_halo(const float* src, void* dst, int cout, int cout_pad, int cin, int cin_pad, int prec, hipStream_t st, int bn = 0);
// Linear-layer packing (CONV_FC): src [cout][cin] fp32 -> bf16 MFMA-fragment order [cout_pad/16][kpad/32][64][8]
// conv_pw_x3.hip: the split precision's pointwise conv and Linear kernels (fragment-ordered hi | lo weight blocks)
bool pw_x3_applicable(int kh, int kw, int stride, int pad, int res_mode, const TView& in, const TView& out);
hipError_t launch_conv_pw_x3(const ConvArgs& a, hipStream_t st);
bool fc_x3_applicable(int kh, int kw, int stride, const TView& in, const TView& out);
hipError_t launch_fc_x3(const ConvArgs& a, hipStream_t st);
hipError_t launch_pack_weights_fcx3(const float* src, void* dst, int cout, int cout_pad, int cin, int kpad, hipStream_t st);
hipError_t launch_pack_weights_fc(const float* src, void* dst, int cout, int cout_pad, int cin, int kpad, int prec, hipStream_t st);
// dw_attn.hip: depth-wise k x k conv (k = 3 | 7, stride 1 | 2, pad k/2; weights fp32 [k*k][C], bias fp32 [C]; residual: RES_AFTER_ACT only)
// fuse_ops.hip: EfficientDet's element-wise operators (squeeze-and-excitation gate, channel scale, BiFPN weighted sum)
bool se_gate_supported(const TView& in, const TView& gate, int cr, uint64_t w_elems, uint64_t b_elems);
// act_hidden: ACT_SILU (0 means SiLU too: files written before the field existed) | ACT_RELU; act_gate: 0 = sigmoid | ACT_HSIGMOID
hipError_t launch_se_gate(const TView& in, const TView& gate, const float* w1, const float* w2, int cr, int act_hidden, int act_gate, int n, int prec, hipStream_t st,
                          const TView* scratch = nullptr);   // scratch: fp32 1x1x(P*C) per frame -> P pixel ranges summed by their own launch
bool scale_supported(const TView& in, const TView& gate, const TView& out);
hipError_t launch_scale(const TView& in, const TView& gate, const TView& out, int n, int prec, hipStream_t st);
bool shuffle_supported(const TView& in, const TView& out, int groups);
hipError_t launch_shuffle(const TView& in, const TView& out, int groups, int n, int prec, hipStream_t st);
bool wsum_supported(int n_in, const TView* ins, const TView& out);
hipError_t launch_wsum(int n_in, const TView* ins, const float* w, const TView& out, int n, int act, int prec, hipStream_t st);
bool dwconv_supported(int k, int stride, int pad, int res_mode, const TView& in, const TView& out);
hipError_t launch_dwconv(const TView& in, const TView& out, const TView& res, int res_mode, const float* wgt, const float* bias, int n, int k,
                         int stride, int pad, int act, int prec, hipStream_t st);
// dw_attn.hip: PSA softmax attention over the H*W tokens of one qkv tensor (per head: key_dim q, key_dim k, head_dim v channels)
bool attention_supported(int nh, int kd, int hd, const TView& qkv, const TView& out);
hipError_t launch_attention(const TView& qkv, const TView& out, int n, int nh, int kd, int hd, float scale, int prec, hipStream_t st);
// NHWC (compute type or fp32) activation view -> NCHW fp32 (debug / parity tap)
hipError_t launch_nhwc_to_nchw(TView in, float* out, int n, int prec, hipStream_t st);

}  // namespace adas
