// engine.h -- in-memory form of a loaded ADASHIP1 model (shared by engine.cpp and pipeline.cpp).
// File structs mirror the struct formats in vehicle-cv-adas_amd/models.py (little-endian, naturally aligned).
#pragma once
#include "common.h"
#include "kernels.h"
#include "conv_ml.h"
#include <map>
#include <string>
#include <vector>

namespace adas {

enum { OP_INPUT = 0, OP_CONV, OP_MAXPOOL, OP_UPSAMPLE2, OP_DETECT_V8, OP_DETECT_V5, OP_LAYERNORM, OP_DWCONV /* depth-wise conv: weights
       [C][kh][kw] in the container, [kh*kw][C] fp32 on the device */, OP_ATTENTION /* params: heads, key_dim, head_dim, scale */, OP_AVGPOOL /* kh x kh, stride, pad: count_include_pad average */,
       OP_DEPTH2SPACE /* (H, W, 4C) -> (2H, 2W, C), channel blocks ordered (dy, dx) */, OP_DETECT_V6 /* params: nc, A, strides; inputs (reg, cls) per level */,
       OP_SE_GATE /* squeeze-and-excitation gate (fuse_ops.hip): params[0] = squeeze width; w = [W1 | b1], b = [W2 | b2]; out: 1x1xC fp32 */,
       OP_SCALE /* inputs (x, gate): x * gate[n][c] */, OP_WSUM /* act(sum_i params[i] * in_i), 2-3 inputs, half-resolution inputs upsampled on the fly */,
       OP_SHUFFLE /* torch channel_shuffle, params[0] = groups: out[j * g + i] = in[i * (C / g) + j] */ };

struct FileHeader {
    char magic[8];
    uint32_t version, n_bufs, n_ops, n_outputs, in_c, in_h, in_w, in_cpad;  // in_cpad: low 16 bits = 8; bit 16 = the source model's I/O was float16
    uint64_t weights_off, weights_bytes;
    double flops;
    char name[64];
};
static_assert(sizeof(FileHeader) == 128, "FileHeader layout");
struct FileBuf {
    uint32_t h, w, c, flags;
};
struct FileOp {
    uint32_t type, n_in;
    int32_t in_buf[8], in_coff[8], in_c[8];
    int32_t out_buf, out_coff, out_c;
    uint32_t kh, kw, stride, pad, act, res_mode;
    int32_t res_buf, res_coff;
    uint32_t flags, reserved, reserved2;
    uint64_t w_off, w_elems, b_off, b_elems;
    double flops;
    float params[8];
    char name[48];
};
static_assert(sizeof(FileOp) == 280, "FileOp layout");
struct FileOut {
    uint32_t buf, offset, ndim, dims[4];
    char name[32];
    uint32_t pad;
};
static_assert(sizeof(FileOut) == 64, "FileOut layout");

struct EngBuf {
    int h, w, c;
    bool f32;
    void* d;
    int alias_of = -1;  // >= 0: shares the device memory of that buffer (FileBuf.flags bit 1, target in flags >> 8)
};
struct EngOp {
    FileOp f;
    std::string name;
    size_t w_off, b_off;  // into the packed device weight arena
    int k, kpad, cout_pad, cin_pad;
    int kernel = 0;  // CONV_* (kernels.h): fixes the weight packing
    bool skip = false;       // fused into a neighbouring launch (input conversion / stem max-pool)
    int fuse_pool = -1;      // CONV_STEM: index of the max-pool op folded into this conv, or -1
    int fuse_conv2 = -1;     // CONV_STEM: index of the 3x3 s2 16->32 conv folded into this launch (YOLO stems), or -1
    int ds_src = -1;         // 3x3 conv whose residual is a 1x1 stride-2 projection: index of that projection conv (folded into this launch
                             // at batches where conv_halo8 takes the layer), or -1
    int ds_user = -1;        // the projection conv's side of the same link
    size_t ds_w_off = 0;     // projection weights re-packed as per-step tiles for the fold
    size_t x3h8_w_off = 0;   // split precision: second packing of a 3x3 s1 conv for conv_halo8_x3.hip (has_x3h8)
    bool has_x3h8 = false;
    int halo_bn = 0;         // CONV_HALO: output channels per workgroup the weights are packed for (0: halo_bn(cout)); plan_halo_bn
    int up_src = -1;         // OP_CONV (1x1): index of the upsample op folded into this conv's activation loads, or -1
    int pool3[2] = {-1, -1}; // OP_MAXPOOL: the two pools chained behind this one, folded into its launch (SPPF), or -1
    int pair_b = -1;         // CONV_PAIR: index of the second conv of the pair this op launches (its own output is never written), or -1
    int c2f[3] = {-1, -1, -1};   // cv1 of a fused C2f block (conv_c2f.hip): indices of the Bottleneck's conv A, conv B and of cv2 (all skipped), or -1
    int det_src[6] = {-1, -1, -1, -1, -1, -1};  // OP_DETECT_V8: the six 1x1 convs (cv2.i.2, cv3.i.2) folded into the decode launch, or -1;
                                                // OP_DETECT_V5: the three per-level 1x1 convs (det_src[0..2])
};
// One multi-layer launch (conv_ml.hip): the non-skipped ops of [first, last] are all convs with a tile body there and run as ONE launch at
// `first`'s position; decided per batch size (the kernels a layer resolves to, and with them its eligibility, depend on the batch).
struct MlSeg {
    int first = 0, last = 0, n_layers = 0, n_items = 0;
    MlPlan* plan = nullptr;
    std::vector<int> ops;     // the member ops, in launch order
};
// A run of consecutive 3x3 halo convs re-ordered by dependency level, the layers of a level launched together (conv_ml.hip grouped launch:
// independent layers, no synchronisation inside the launch).  Decided per batch size like the multi-layer launches.
struct GroupStep {
    MlGroup* group = nullptr;    // >= 2 independent layers in one launch, or
    int op = -1;                 // one layer on its own kernel
    std::vector<int> members;
};
struct GroupRun {
    int first = 0, last = 0;
    std::vector<GroupStep> steps;
    std::vector<int> ops;
};
struct EngOut {
    uint32_t buf, offset, ndim, dims[4];
    size_t elems;  // per frame
    std::string name;
};

// packed_in: d_in is the (c0,c1,c2,0) bf16 NHWC tensor of adas_preprocess_*_packed (fused first layer only)
int engine_run_op(struct ::adas_engine* e, int i, const float* d_in, int batch, hipStream_t st, bool packed_in = false);
int engine_forward(struct ::adas_engine* e, const float* d_in, int batch, hipStream_t st, bool packed_in = false);
int engine_prepare(struct ::adas_engine* e, int batch);   // multi-layer launch tables of this batch size (never inside a stream capture)

}  // namespace adas

struct adas_engine {
    int prec = 0, max_batch = 1;
    adas::FileHeader hdr;
    std::string name;
    std::vector<adas::EngBuf> bufs;
    std::vector<adas::EngOp> ops;
    std::vector<adas::EngOut> outs;
    void* d_weights = nullptr;
    float* d_input = nullptr;
    size_t weight_bytes = 0, act_bytes = 0;
    std::vector<hipEvent_t> events;
    std::vector<hipEvent_t> step_events;   // adas_engine_profile: one per step of a grouped run
    hipStream_t last = 0;
    std::map<int, std::vector<adas::MlSeg>> ml;   // batch -> multi-layer launches (adas_engine_prepare); absent: not prepared, per-layer launches
    std::map<int, std::vector<adas::GroupRun>> groups;   // batch -> grouped launches of independent layers (default path)
    bool group_on = false;                         // ADAS_NO_GROUP=1 at creation keeps every layer its own launch
    bool ml_on = false;                            // multi-layer launches enabled for this engine (read from the environment at creation)
    std::vector<char> buf_aliased;                 // buffer takes part in an alias (Graph.alias): stays out of multi-layer launches
    float* sink_conf = nullptr;   // adas_engine_set_detect_sink: the fused v8 Detect writes per-anchor (best probability, class) here
    int* sink_cls = nullptr;      // instead of the head's class rows (pipeline steps)
};
