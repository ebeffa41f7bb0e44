// post_kernels.hip -- GPU-resident post-processing kernels + their C-ABI entry points.
//   yolo_scan_v8 / yolo_scan_v5 : per-anchor class argmax + confidence  (yoloDetector.py:120-127)
//   yolo_post_kernel            : compaction -> box -> inverse letterbox -> NMS -> RectInfo gather
//   ufld_decode_kernel          : row/col-anchor lane decode           (ultrafastLaneDetectorV2.py:114-181)
//   bytetrack_*_kernel          : BYTETracker.update / reset           (byteTracker.py:62-200)
// Roofline: all HBM-bound by construction (2.82 MB fp32 head per v8 frame, 8.57 MB per v5 frame,
// 365 KB of lane logits); at these sizes the honest KPI is us per frame.
#include "common.h"
#ifdef ADAS_YP_PROF
__device__ unsigned long long g_yp_prof[8];
extern "C" int adas_debug_yolo_prof(unsigned long long* out8) {
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_yp_prof), 64) == hipSuccess ? 0 : -1;
}
#endif
#include "post_core.h"
#include "track_core.h"
#include "lane_core.h"
#include <new>
#include <string.h>
#include <vector>

using namespace adas;

// -------------------------------------------------------------------------------------
// scan, v8-family layout [4+nc][A]: wave w of the block owns class group w (ascending class
// ranges), lanes own 4 consecutive anchors (one 16-byte load per class row -> 1 KiB per wave).
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void yolo_scan_v8(const float* __restrict__ head, int A, int nc,
                                                    float* __restrict__ best_conf, int* __restrict__ best_cls) {
    __shared__ float s_v[4][64][4];
    __shared__ int s_i[4][64][4];
    const int frame = blockIdx.y;
    head += (size_t)frame * (size_t)(4 + nc) * A;
    best_conf += (size_t)frame * A;
    best_cls += (size_t)frame * A;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int a0 = (blockIdx.x * 64 + lane) * 4;
    const int cpg = (nc + 3) / 4;
    const int c_lo = grp * cpg, c_hi = min(nc, c_lo + cpg);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    int bi[4] = {-1, -1, -1, -1};
    const bool vec = (A & 3) == 0;
    if (a0 < A) {
        for (int c = c_lo; c < c_hi; ++c) {
            const float* row = head + (size_t)(4 + c) * A + a0;
            float v[4];
            if (vec) {
                float4 q = *reinterpret_cast<const float4*>(row);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = (a0 + t < A) ? row[t] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (bi[t] < 0 || v[t] > bv[t]) {
                    bv[t] = v[t];
                    bi[t] = c;
                }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        s_v[grp][lane][t] = bv[t];
        s_i[grp][lane][t] = bi[t];
    }
    __syncthreads();
    if (grp == 0 && a0 < A) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = bv[t];
            int i = bi[t];
            for (int g = 1; g < 4; ++g) {
                int oi = s_i[g][lane][t];
                float ov = s_v[g][lane][t];
                if (oi >= 0 && (i < 0 || ov > v)) {  // later groups hold higher class ids: strict > keeps the first max
                    v = ov;
                    i = oi;
                }
            }
            if (a0 + t < A) {
                best_conf[a0 + t] = v;
                best_cls[a0 + t] = i < 0 ? 0 : i;
            }
        }
    }
}

// scan, v5-family layout [A][5+nc]: 16 lanes per row, four rows per wave pass (a wave per row left 12 dependent load -> 6-step reduce
// round trips per wave: 1.5 TB/s on the 8.57 MB head); conf = cls * obj rounded in fp32, first maximum wins (np.argmax).
__global__ __launch_bounds__(256) void yolo_scan_v5(const float* __restrict__ head, int A, int nc,
                                                    float* __restrict__ best_conf, int* __restrict__ best_cls) {
    const int frame = blockIdx.y;
    const int no = 5 + nc;
    head += (size_t)frame * (size_t)A * no;
    best_conf += (size_t)frame * A;
    best_cls += (size_t)frame * A;
    const int lane = threadIdx.x & 63, sub = lane & 15, q = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    for (int a0 = wave * 4; a0 < A; a0 += nwaves * 4) {
        const int a = a0 + q;
        const bool ok = a < A;
        const float* row = head + (size_t)(ok ? a : 0) * no;
        const float obj = row[4];
        float bv = 0.f;
        int bi = 0x7fffffff;
        for (int c = sub; c < nc; c += 16) {
            float p = row[5 + c] * obj;
            if (bi == 0x7fffffff || p > bv) {
                bv = p;
                bi = c;
            }
        }
        for (int off = 8; off > 0; off >>= 1) {
            float ov = __shfl_down(bv, off, 16);
            int oi = __shfl_down(bi, off, 16);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) {
                bv = ov;
                bi = oi;
            }
        }
        if (sub == 0 && ok) {
            best_conf[a] = bv;
            best_cls[a] = bi == 0x7fffffff ? 0 : bi;
        }
    }
}

struct YoloPostDev {
    YoloPostCfg cfg;
    const float* head;
    size_t head_stride;
    float* best_conf;
    int* best_cls;
    int* counts;
    int* cand_anchor;
    double* cand_xywh;
    double* cand_conf;
    int* cand_cls;
    int* keep;
    double* det_xywh;
    double* det_conf;
    int* det_cls;
    int* det_xyxy_i;
    double* det_xyxy_d;
    unsigned char* spill;   // candidate capacities past the LDS arena (> 2048): the frame's working set lives here (HBM), else nullptr
    size_t spill_stride;
};

// SPILL: the NMS working set (YoloLds: coordinates, scores, areas, order, reduction scratch) of a frame is carved out of a per-frame
// HBM workspace instead of LDS.  The reference appends candidates without limit (yoloDetector.py:126-133); a post-processor created
// with max_candidates above what LDS holds keeps that semantics at HBM latency (same code, same arithmetic, same results).
template <bool SPILL>
__global__ __launch_bounds__(256) void yolo_post_kernel(YoloPostDev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_lds[];
    const int b = blockIdx.x;
    unsigned char* smem = SPILL ? d.spill + (size_t)b * d.spill_stride : smem_lds;
    const size_t cap = d.cfg.cap;
    YoloPostFrame f;
    f.head = d.head + b * d.head_stride;
    f.best_conf = d.best_conf + (size_t)b * d.cfg.A;
    f.best_cls = d.best_cls + (size_t)b * d.cfg.A;
    f.counts = d.counts + b * 4;
    f.cand_anchor = d.cand_anchor + b * cap;
    f.cand_xywh = d.cand_xywh + b * cap * 4;
    f.cand_conf = d.cand_conf + b * cap;
    f.cand_cls = d.cand_cls + b * cap;
    f.keep = d.keep + b * cap;
    f.det_xywh = d.det_xywh + b * cap * 4;
    f.det_conf = d.det_conf + b * cap;
    f.det_cls = d.det_cls + b * cap;
    f.det_xyxy_i = d.det_xyxy_i + b * cap * 4;
    f.det_xyxy_d = d.det_xyxy_d + b * cap * 4;
    Ctx c{(int)threadIdx.x, (int)blockDim.x};
    yolo_post_frame(c, d.cfg, f, smem);
}

// -------------------------------------------------------------------------------------
struct UfldDev {
    UfldCfg cfg;
    const float *loc_row, *loc_col, *exist_row, *exist_col;
    size_t s_lr, s_lc, s_er, s_ec;
    int *lane_cnt, *lane_det, *lane_pts;
};

__global__ __launch_bounds__(640) void ufld_decode_kernel(UfldDev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    UfldFrame f;
    f.loc_row = d.loc_row + b * d.s_lr;
    f.loc_col = d.loc_col + b * d.s_lc;
    f.exist_row = d.exist_row + b * d.s_er;
    f.exist_col = d.exist_col + b * d.s_ec;
    f.lane_cnt = d.lane_cnt + b * 4;
    f.lane_det = d.lane_det + b * 4;
    f.lane_pts = d.lane_pts + (size_t)b * 4 * ADAS_UFLD_MAXPTS * 2;
    Ctx c{(int)threadIdx.x, (int)blockDim.x};
    ufld_decode_frame(c, d.cfg, f, smem);
}

struct Ufld1Dev {
    Ufld1Cfg cfg;
    const float* out;
    size_t stride;
    int *lane_cnt, *lane_det, *lane_pts;
};

__global__ __launch_bounds__(256) void ufld1_decode_kernel(Ufld1Dev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    Ctx c{(int)threadIdx.x, (int)blockDim.x};
    ufld1_decode_frame(c, d.cfg, d.out + b * d.stride, d.lane_cnt + b * 4, d.lane_det + b * 4,
                       d.lane_pts + (size_t)b * 4 * ADAS_UFLD_MAXPTS * 2, smem);
}

struct LaneGeomDev {
    LaneGeomCfg cfg;
    const int *lane_cnt, *lane_det, *lane_pts;  // the decoder's arrays
    int* hdr;       // [B][8]
    double* vals;   // [B][2]
    int* area;      // [B][2*img_h][2]
    int* bird;      // [B][4][MAXPTS][2]
    double* fx;     // [B][2*img_h]
    int* idx;       // [B][2*img_h]
};

__global__ __launch_bounds__(256) void lane_geometry_kernel(LaneGeomDev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const size_t b = blockIdx.x, H = d.cfg.img_h;
    LaneGeomFrame f;
    f.lane_cnt = d.lane_cnt + b * 4;
    f.lane_det = d.lane_det + b * 4;
    f.lane_pts = d.lane_pts + b * 4 * ADAS_LANE_MAXPTS * 2;
    f.hdr = d.hdr + b * 8;
    f.vals = d.vals + b * 2;
    f.area = d.area + b * 4 * H;
    f.bird = d.bird + b * 4 * ADAS_LANE_MAXPTS * 2;
    f.fx = d.fx + b * 2 * H;
    f.idx = d.idx + b * 2 * H;
    Ctx c{(int)threadIdx.x, (int)blockDim.x};
    lane_geometry_frame(c, d.cfg, f, smem);
}

// -------------------------------------------------------------------------------------
struct BtDev {
    BtParams P;
    unsigned char* base;  // per-stream regions
    size_t stream_bytes;
    const double* xyxy;
    const double* score;
    const int* cls;
    const int* counts;
    int det_stride, count_stride, count_index, first_stream;
    int n_frames = 1, frame_slabs = 0;   // temporal micro-batch: this launch consumes n_frames consecutive frames of every stream, frame f of
                                         // the stream in block b at detection slab f * frame_slabs + b
    unsigned char* snap = nullptr;       // micro-batch: [frame][stream] copies of (header, out list) after each frame's update -- the message
    size_t snap_bytes = 0;               // BYTETracker.update returns every frame (byteTracker.py:185); the live state only holds the last one
};

__host__ __device__ inline size_t bt_align(size_t x) { return (x + 63) & ~(size_t)63; }
__host__ __device__ inline size_t bt_snap_bytes(int MT) { return bt_align(sizeof(BtHeader)) + bt_align((size_t)2 * MT * sizeof(BtOut)); }
__host__ __device__ inline size_t bt_stream_bytes(int MT, int MD) {
    return bt_align(sizeof(BtHeader)) + bt_align((size_t)MT * sizeof(BtTrack)) + bt_align((size_t)MT * MD * 8) +
           bt_align((size_t)2 * MT * sizeof(BtOut)) + 2 * bt_align((size_t)MT * 4) + bt_align((size_t)MT * ADAS_BT_TRAJ * 32);
}
__host__ __device__ inline BtStream bt_view(unsigned char* p, int MT, int MD) {
    BtStream S;
    S.hdr = (BtHeader*)p; p += bt_align(sizeof(BtHeader));
    S.slots = (BtTrack*)p; p += bt_align((size_t)MT * sizeof(BtTrack));
    S.cost = (double*)p; p += bt_align((size_t)MT * MD * 8);
    S.out = (BtOut*)p; p += bt_align((size_t)2 * MT * sizeof(BtOut));
    S.tracked = (int*)p; p += bt_align((size_t)MT * 4);
    S.lost = (int*)p; p += bt_align((size_t)MT * 4);
    S.traj = (double*)p;
    return S;
}

__global__ __launch_bounds__(256) void bytetrack_update_kernel(BtDev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = d.first_stream + blockIdx.x;
    BtStream S = bt_view(d.base + (size_t)s * d.stream_bytes, d.P.MT, d.P.MD);
    Ctx c{(int)threadIdx.x, (int)blockDim.x};
    for (int f = 0; f < d.n_frames; ++f) {   // one update per frame, in temporal order (BYTETracker.update is called once per frame)
        const int q = f * d.frame_slabs + (int)blockIdx.x;  // detection slab index
        BtDet det;
        det.tlbr = d.xyxy + (size_t)q * d.det_stride * 4;
        det.score = d.score + (size_t)q * d.det_stride;
        det.cls = d.cls + (size_t)q * d.det_stride;
        det.nd = d.counts[(size_t)q * d.count_stride + d.count_index];
        bytetrack_update(c, d.P, S, det, smem);
        __syncthreads();   // the next frame's update starts from the track table and LDS this one leaves behind
        if (d.snap) {      // this frame's message: header + tracked / lost lists, 8-byte words
            unsigned char* q = d.snap + ((size_t)f * d.frame_slabs + blockIdx.x) * d.snap_bytes;
            const unsigned long long* sh = (const unsigned long long*)S.hdr;
            const unsigned long long* so = (const unsigned long long*)S.out;
            unsigned long long* dh = (unsigned long long*)q;
            unsigned long long* dout = (unsigned long long*)(q + bt_align(sizeof(BtHeader)));
            const int nw = (S.hdr->n_tracked + S.hdr->n_lost) * (int)(sizeof(BtOut) / 8);
            for (int i = threadIdx.x; i < (int)(sizeof(BtHeader) / 8); i += blockDim.x) dh[i] = sh[i];
            for (int i = threadIdx.x; i < nw; i += blockDim.x) dout[i] = so[i];
            __syncthreads();
        }
    }
}

// one stream's trajectories in message order into the handle's gather buffer: [MT] lengths, then [MT][30][4] boxes
__global__ __launch_bounds__(256) void bytetrack_traj_kernel(BtDev d, int stream, int* lens, double* out) {
    BtStream S = bt_view(d.base + (size_t)stream * d.stream_bytes, d.P.MT, d.P.MD);
    Ctx c{(int)threadIdx.x, (int)blockDim.x};
    bytetrack_gather_trajectories(c, S, lens, out);
}

__global__ __launch_bounds__(256) void bytetrack_reset_kernel(BtDev d) {
    const int s = d.first_stream + blockIdx.x;
    BtStream S = bt_view(d.base + (size_t)s * d.stream_bytes, d.P.MT, d.P.MD);
    Ctx c{(int)threadIdx.x, (int)blockDim.x};
    bytetrack_reset(c, d.P, S);
}

// =====================================================================================
// C ABI
// =====================================================================================
constexpr int LGEO_MSG_INTS = 8 + 4 + 4 * ADAS_LANE_MAXPTS * 2;      // header, two doubles (as 4 ints), bird points
__global__ __launch_bounds__(256) void lane_geometry_pack_kernel(const int* hdr, const double* vals, const int* bird, int frame, int* msg) {
    const int t = threadIdx.x;
    if (t < 8) msg[t] = hdr[(size_t)frame * 8 + t];
    if (t < 4) msg[8 + t] = reinterpret_cast<const int*>(vals + (size_t)frame * 2)[t];
    for (int i = t; i < 4 * ADAS_LANE_MAXPTS * 2; i += blockDim.x) msg[12 + i] = bird[(size_t)frame * 4 * ADAS_LANE_MAXPTS * 2 + i];
}
constexpr int UFLD_MSG_INTS = 8 + 4 * ADAS_UFLD_MAXPTS * 2;
__global__ __launch_bounds__(256) void ufld_pack_kernel(const int* cnt, const int* det, const int* pts, int frame, int* msg) {
    const int t = threadIdx.x;
    if (t < 4) {
        msg[t] = cnt[frame * 4 + t];
        msg[4 + t] = det[frame * 4 + t];
    }
    for (int i = t; i < 4 * ADAS_UFLD_MAXPTS * 2; i += blockDim.x) msg[8 + i] = pts[(size_t)frame * 4 * ADAS_UFLD_MAXPTS * 2 + i];
}

// The survivors of ONE frame as one message: [n_found, n_candidates, n_keep, flags] then n_keep records of 64 bytes
// {double xywh[4]; double conf; int cls; int keep; int xyxy[4]} -- what YoloDetector.DetectFrame reads after every frame, in one
// device-to-host copy instead of one per array (adas_yolo_post_fetch: up to ten).
constexpr size_t YOLO_MSG_HDR = 16, YOLO_MSG_REC = 64, YOLO_MSG_FIRST = 62;   // 16 + 62 * 64 = 3984 B: the first copy; more survivors -> a second one
__global__ __launch_bounds__(256) void yolo_pack_kernel(YoloPostDev d, int frame, unsigned char* msg) {
    const size_t cap = d.cfg.cap, b = frame;
    const int* cnt = d.counts + b * 4;
    if (threadIdx.x < 4) ((int*)msg)[threadIdx.x] = cnt[threadIdx.x];
    const int k = cnt[2];
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        unsigned char* r = msg + YOLO_MSG_HDR + (size_t)i * YOLO_MSG_REC;
        double* f = (double*)r;
        int* w = (int*)(r + 40);
        for (int j = 0; j < 4; ++j) f[j] = d.det_xywh[(b * cap + i) * 4 + j];
        f[4] = d.det_conf[b * cap + i];
        w[0] = d.det_cls[b * cap + i];
        w[1] = d.keep[b * cap + i];
        for (int j = 0; j < 4; ++j) w[2 + j] = d.det_xyxy_i[(b * cap + i) * 4 + j];
    }
}

struct adas_yolo_post {
    adas_yolo_post_params p;
    int max_batch;
    YoloPostDev dev;
    void* arena;
    unsigned char* msg;      // the packed survivor message of adas_yolo_post_fetch_dets (device) ...
    unsigned char* h_msg;    // ... and its host landing buffer
    hipStream_t last;
};
struct adas_ufld_decode {
    adas_ufld_params p;
    int max_batch;
    int v1;  // created by adas_ufld1_decode_create
    UfldDev dev;
    Ufld1Dev dev1;
    void* arena;
    int* msg = nullptr;     // one frame's lanes as one message: [4 counts][4 detected][4 x MAXPTS x 2 points] (device) ...
    int* h_msg = nullptr;   // ... and its pinned landing buffer (adas_ufld_decode_fetch: one copy instead of three)
    hipStream_t last;
};
struct EffdetDev {
    EffdetCfg cfg;
    const float* boxes;
    const int* ids;
    const float* confs;
    const int* counts_in;  // [B] device copy of the caller's per-frame counts
    int* count;            // [B]
    float* xywh;           // [B][cap][4]
    float* conf;           // [B][cap]
    int* cls;              // [B][cap]
    int* xyxy_i;           // [B][cap][4]
};
__global__ __launch_bounds__(256) void effdet_post_kernel(EffdetDev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const size_t b = blockIdx.x, cap = d.cfg.cap;
    EffdetFrame f{d.boxes + b * cap * 4, d.ids + b * cap, d.confs + b * cap, d.counts_in[b], d.count + b,
                  d.xywh + b * cap * 4, d.conf + b * cap, d.cls + b * cap, d.xyxy_i + b * cap * 4};
    Ctx c{(int)threadIdx.x, (int)blockDim.x};
    effdet_post_frame(c, d.cfg, f, (int*)smem);
}
struct adas_effdet_post {
    adas_effdet_post_params p;
    int max_batch;
    EffdetDev dev;
    int* d_counts_in;
    void* arena;
    hipStream_t last;
};
struct EffdetTailDev {
    EffdetTailCfg cfg;
    const float* reg[5];
    const float* cls[5];
    size_t rows[5];        // cells_l * 9: frame stride of level l in rows
    int* count;            // [B][2]
    float* boxes;          // [B][max_det][4]
    int* ids;              // [B][max_det]
    float* confs;          // [B][max_det]
};
__global__ __launch_bounds__(1024) void effdet_tail_kernel(EffdetTailDev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const size_t b = blockIdx.x, md = d.cfg.max_det;
    EffdetTailFrame f;
    for (int l = 0; l < 5; ++l) {
        f.reg[l] = d.reg[l] + b * d.rows[l] * 4;
        f.cls[l] = d.cls[l] + b * d.rows[l] * d.cfg.nc;
    }
    f.count = d.count + b * 2; f.boxes = d.boxes + b * md * 4; f.ids = d.ids + b * md; f.confs = d.confs + b * md;
    Ctx c{(int)threadIdx.x, (int)blockDim.x};
    effdet_tail_frame(c, d.cfg, f, smem);
}
struct adas_effdet_tail {
    adas_effdet_tail_params p;
    int max_batch;
    EffdetTailDev dev;
    void* arena;
    hipStream_t last;
};
struct adas_lane_geometry {
    int max_batch;
    LaneGeomDev dev;
    void* arena;
    int* msg = nullptr;     // one frame's fixed-size results as one message: [8 header ints][2 doubles][4 x MAXPTS x 2 bird points] (device) ...
    int* h_msg = nullptr;   // ... and its pinned landing buffer (adas_lane_geometry_fetch: one copy + the area polygon instead of four)
    hipStream_t last;
};
struct adas_bytetrack {
    adas_bytetrack_params p;
    int n_streams;
    BtDev dev;
    void* arena;
    double* h_stage_d;  // device staging for update_host
    hipStream_t last;
    void* snap = nullptr;      // per-frame snapshots of the last update_device_frames launch ([n_frames][n_streams], grown on demand)
    int snap_frames = 0, snap_streams = 0;
    int last_frames = 0;   // frames of the LAST adas_bytetrack_update_device_frames launch (0: a single-frame launch, no per-frame snapshots)
};

namespace adas {
int handle_max_batch(const ::adas_yolo_post* h) { return h ? h->max_batch : 0; }
int handle_max_batch(const ::adas_ufld_decode* h) { return h ? h->max_batch : 0; }
int handle_max_batch(const ::adas_lane_geometry* h) { return h ? h->max_batch : 0; }
}  // namespace adas

template <class T>
static T* carve(unsigned char*& p, size_t n) {
    T* r = (T*)p;
    p += (n * sizeof(T) + 255) & ~(size_t)255;
    return r;
}

extern "C" {

int adas_letterbox_params(int src_h, int src_w, int dst_h, int dst_w, int keep_ratio, adas_yolo_post_params* p) {
    ADAS_REQUIRE(p && src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0, ADAS_ERR_INVALID, "adas_letterbox_params: bad argument");
    int padh = 0, padw = 0, newh = dst_h, neww = dst_w;  // utils.py:43-52
    if (keep_ratio && src_h != src_w) {
        double hw = (double)src_h / (double)src_w;
        if (hw > 1) {
            neww = (int)((double)dst_w / hw);
            padw = (int)((double)(dst_w - neww) * 0.5);
        } else {
            newh = (int)((double)dst_h * hw) + 1;
            padh = (int)((double)(dst_h - newh) * 0.5);
        }
    }
    p->pad_h = padh;
    p->pad_w = padw;
    p->ratio_h = (double)src_h / (double)newh;  // utils.py:65-68
    p->ratio_w = (double)src_w / (double)neww;
    return ADAS_OK;
}

int adas_yolo_post_create(const adas_yolo_post_params* p, int max_batch, adas_yolo_post** out) {
    ADAS_REQUIRE(p && out && max_batch > 0, ADAS_ERR_INVALID, "adas_yolo_post_create: bad argument");
    ADAS_REQUIRE(p->layout == ADAS_HEAD_V8 || p->layout == ADAS_HEAD_V5 || p->layout == ADAS_HEAD_V5_LITE, ADAS_ERR_INVALID,
                 "unknown head layout %d", p->layout);
    ADAS_REQUIRE(p->nms_mode == ADAS_NMS_REFERENCE || p->nms_mode == ADAS_NMS_GREEDY, ADAS_ERR_INVALID, "unknown nms mode %d", p->nms_mode);
    ADAS_REQUIRE(p->num_anchors > 0 && p->num_classes > 0, ADAS_ERR_INVALID, "bad head geometry");
    ADAS_REQUIRE(p->box_score >= 0.0, ADAS_ERR_INVALID, "box_score must be >= 0");
    ADAS_REQUIRE(p->max_candidates >= 16 && p->max_candidates <= (p->num_anchors > 2048 ? p->num_anchors : 2048), ADAS_ERR_INVALID,
                 "max_candidates must be in [16, max(2048, num_anchors)]");
    ADAS_REQUIRE(adas_device_count() > 0, ADAS_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    adas_yolo_post* h = new (std::nothrow) adas_yolo_post();
    ADAS_REQUIRE(h, ADAS_ERR_INVALID, "out of host memory");
    h->p = *p;
    h->max_batch = max_batch;
    h->last = 0;
    const size_t B = max_batch, A = p->num_anchors, cap = p->max_candidates;
    const bool spill = p->max_candidates > 2048;                       // past the LDS arena: per-frame HBM workspace
    const size_t spill_stride = spill ? ((YoloLds::bytes(p->max_candidates, 256) + 255) & ~(size_t)255) : 0;
    size_t bytes = B * (A * 8 + 16 + cap * (4 + 32 + 8 + 4 + 4 + 32 + 8 + 4 + 16 + 32)) + 17 * 256 + B * spill_stride + 256 +
                   YOLO_MSG_HDR + cap * YOLO_MSG_REC;
    h->h_msg = nullptr;
    if (hipHostMalloc((void**)&h->h_msg, YOLO_MSG_HDR + cap * YOLO_MSG_REC, hipHostMallocDefault) != hipSuccess || hipMalloc(&h->arena, bytes) != hipSuccess) {
        if (h->h_msg) hipHostFree(h->h_msg);
        delete h;
        return hip_fail(hipGetLastError(), "hipMalloc(yolo_post arena)", __FILE__, __LINE__);
    }
    hipMemset(h->arena, 0, bytes);
    unsigned char* q = (unsigned char*)h->arena;
    YoloPostDev& d = h->dev;
    d.cfg = YoloPostCfg{p->layout, p->num_anchors, p->num_classes, p->box_score, p->iou_thr, p->nms_mode,
                        p->pad_h, p->pad_w, p->ratio_h, p->ratio_w, p->max_candidates, 0, 0};
    d.head = nullptr;
    d.head_stride = p->layout == ADAS_HEAD_V8 ? (size_t)(4 + p->num_classes) * A : (size_t)(5 + p->num_classes) * A;
    d.best_conf = carve<float>(q, B * A);
    d.best_cls = carve<int>(q, B * A);
    d.counts = carve<int>(q, B * 4);
    d.cand_anchor = carve<int>(q, B * cap);
    d.cand_xywh = carve<double>(q, B * cap * 4);
    d.cand_conf = carve<double>(q, B * cap);
    d.cand_cls = carve<int>(q, B * cap);
    d.keep = carve<int>(q, B * cap);
    d.det_xywh = carve<double>(q, B * cap * 4);
    d.det_conf = carve<double>(q, B * cap);
    d.det_cls = carve<int>(q, B * cap);
    d.det_xyxy_i = carve<int>(q, B * cap * 4);
    d.det_xyxy_d = carve<double>(q, B * cap * 4);
    d.spill = spill ? carve<unsigned char>(q, B * spill_stride) : nullptr;
    d.spill_stride = spill_stride;
    h->msg = carve<unsigned char>(q, YOLO_MSG_HDR + cap * YOLO_MSG_REC);
    size_t lds = spill ? 0 : YoloLds::bytes(p->max_candidates, 256);
    if (!spill && hipFuncSetAttribute((const void*)yolo_post_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        hipFree(h->arena);
        hipHostFree(h->h_msg);
        delete h;
        return hip_fail(hipGetLastError(), "hipFuncSetAttribute(yolo_post_kernel)", __FILE__, __LINE__);
    }
    *out = h;
    return ADAS_OK;
}

int adas_yolo_post_destroy(adas_yolo_post* h) {
    if (!h) return ADAS_OK;
    hipFree(h->arena);
    hipHostFree(h->h_msg);
    delete h;
    return ADAS_OK;
}

int adas_yolo_post_head_shape(const adas_yolo_post* h, int32_t* layout, int32_t* num_anchors, int32_t* num_classes) {
    ADAS_REQUIRE(h, ADAS_ERR_INVALID, "adas_yolo_post_head_shape: null handle");
    if (layout) *layout = h->p.layout;
    if (num_anchors) *num_anchors = h->p.num_anchors;
    if (num_classes) *num_classes = h->p.num_classes;
    return ADAS_OK;
}

int adas_yolo_post_set_input_size(adas_yolo_post* h, int in_h, int in_w) {
    ADAS_REQUIRE(h && in_h >= 32 && in_w >= 32, ADAS_ERR_INVALID, "adas_yolo_post_set_input_size: bad argument");
    if (h->p.layout == ADAS_HEAD_V5_LITE) {
        long rows = 0;
        for (int i = 0; i < 3; ++i) rows += 3L * (in_h / (8 << i)) * (in_w / (8 << i));
        ADAS_REQUIRE(rows <= h->p.num_anchors, ADAS_ERR_INVALID, "a %dx%d input has %ld grid rows but the head has only %d", in_h, in_w,
                     rows, h->p.num_anchors);
    }
    h->dev.cfg.in_h = in_h;
    h->dev.cfg.in_w = in_w;
    adas::bump_config_generation();
    return ADAS_OK;
}

int adas_yolo_post_scan_views(adas_yolo_post* h, float** d_best_conf, int32_t** d_best_cls) {
    ADAS_REQUIRE(h && d_best_conf && d_best_cls, ADAS_ERR_INVALID, "adas_yolo_post_scan_views: bad argument");
    *d_best_conf = h->dev.best_conf;
    *d_best_cls = h->dev.best_cls;
    return ADAS_OK;
}
int adas_yolo_post_run_prescanned(adas_yolo_post* h, const float* d_head, int batch, void* stream) {
    ADAS_REQUIRE(h && d_head && batch > 0 && batch <= h->max_batch, ADAS_ERR_INVALID, "adas_yolo_post_run_prescanned: bad argument (batch %d, max %d)", batch,
                 h ? h->max_batch : 0);
    ADAS_REQUIRE(h->p.layout == ADAS_HEAD_V8 || h->p.layout == ADAS_HEAD_V5, ADAS_ERR_INVALID, "adas_yolo_post_run_prescanned: v8- and v5-layout heads only");
    hipStream_t st = (hipStream_t)stream;
    h->last = st;
    YoloPostDev d = h->dev;
    d.head = d_head;
    if (d.spill) {
        hipLaunchKernelGGL(yolo_post_kernel<true>, dim3(batch), dim3(256), 0, st, d);
    } else {
        size_t lds = YoloLds::bytes(h->p.max_candidates, 256);
        hipLaunchKernelGGL(yolo_post_kernel<false>, dim3(batch), dim3(256), lds, st, d);
    }
    ADAS_HIP_TRY(hipGetLastError());
    return ADAS_OK;
}
int adas_yolo_post_run(adas_yolo_post* h, const float* d_head, int batch, void* stream) {
    ADAS_REQUIRE(h && d_head && batch > 0 && batch <= h->max_batch, ADAS_ERR_INVALID, "adas_yolo_post_run: bad argument (batch %d, max %d)", batch, h ? h->max_batch : 0);
    ADAS_REQUIRE(h->p.layout != ADAS_HEAD_V5_LITE || h->dev.cfg.in_h > 0, ADAS_ERR_INVALID,
                 "v5-lite head: call adas_yolo_post_set_input_size first");
    hipStream_t st = (hipStream_t)stream;
    h->last = st;
    YoloPostDev d = h->dev;
    d.head = d_head;
    const int A = h->p.num_anchors, nc = h->p.num_classes;
    if (h->p.layout == ADAS_HEAD_V8) {
        dim3 grid(((A + 3) / 4 + 63) / 64, batch);
        hipLaunchKernelGGL(yolo_scan_v8, grid, dim3(256), 0, st, d_head, A, nc, d.best_conf, d.best_cls);
    } else {
        dim3 grid(512, batch);
        hipLaunchKernelGGL(yolo_scan_v5, grid, dim3(256), 0, st, d_head, A, nc, d.best_conf, d.best_cls);
    }
    if (d.spill) {
        hipLaunchKernelGGL(yolo_post_kernel<true>, dim3(batch), dim3(256), 0, st, d);
    } else {
        size_t lds = YoloLds::bytes(h->p.max_candidates, 256);
        hipLaunchKernelGGL(yolo_post_kernel<false>, dim3(batch), dim3(256), lds, st, d);
    }
    ADAS_HIP_TRY(hipGetLastError());
    return ADAS_OK;
}

int adas_yolo_post_profile(adas_yolo_post* h, const float* d_head, int batch, int iters, float ms[2]) {
    ADAS_REQUIRE(h && d_head && ms && batch > 0 && batch <= h->max_batch && iters > 0, ADAS_ERR_INVALID, "adas_yolo_post_profile: bad argument");
    ADAS_REQUIRE(h->p.layout != ADAS_HEAD_V5_LITE || h->dev.cfg.in_h > 0, ADAS_ERR_INVALID, "v5-lite head: call adas_yolo_post_set_input_size first");
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    for (auto& e : ev) ADAS_HIP_TRY(hipEventCreate(&e));
    YoloPostDev d = h->dev;
    d.head = d_head;
    h->last = 0;
    const int A = h->p.num_anchors, nc = h->p.num_classes;
    const size_t lds = d.spill ? 0 : YoloLds::bytes(h->p.max_candidates, 256);
    ms[0] = ms[1] = 0.f;
    int rc = ADAS_OK;
    for (int it = 0; it < iters && rc == ADAS_OK; ++it) {
        (void)hipEventRecord(ev[0], 0);
        if (h->p.layout == ADAS_HEAD_V8) hipLaunchKernelGGL(yolo_scan_v8, dim3(((A + 3) / 4 + 63) / 64, batch), dim3(256), 0, 0, d_head, A, nc, d.best_conf, d.best_cls);
        else hipLaunchKernelGGL(yolo_scan_v5, dim3(512, batch), dim3(256), 0, 0, d_head, A, nc, d.best_conf, d.best_cls);
        (void)hipEventRecord(ev[1], 0);
        if (d.spill) hipLaunchKernelGGL(yolo_post_kernel<true>, dim3(batch), dim3(256), 0, 0, d);
        else hipLaunchKernelGGL(yolo_post_kernel<false>, dim3(batch), dim3(256), lds, 0, d);
        (void)hipEventRecord(ev[2], 0);
        hipError_t e = hipEventSynchronize(ev[2]);
        if (e == hipSuccess) e = hipGetLastError();
        float a = 0.f, b = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&a, ev[0], ev[1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&b, ev[1], ev[2]);
        if (e != hipSuccess) rc = hip_fail(e, "adas_yolo_post_profile", __FILE__, __LINE__);
        ms[0] += a / (float)iters;
        ms[1] += b / (float)iters;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

int adas_yolo_post_fetch(adas_yolo_post* h, int frame, adas_yolo_counts* counts, int32_t* cand_anchor, double* cand_xywh,
                         double* cand_conf, int32_t* cand_cls, int32_t* keep, double* det_xywh, double* det_conf,
                         int32_t* det_cls, int32_t* det_xyxy_int) {
    ADAS_REQUIRE(h && frame >= 0 && frame < h->max_batch, ADAS_ERR_INVALID, "adas_yolo_post_fetch: bad frame index");
    ADAS_HIP_TRY(hipStreamSynchronize(h->last));
    const YoloPostDev& d = h->dev;
    const size_t cap = h->p.max_candidates, b = frame;
    int c4[4];
    ADAS_HIP_TRY(hipMemcpy(c4, d.counts + b * 4, 16, hipMemcpyDeviceToHost));
    if (counts) {
        counts->n_found = c4[0];
        counts->n_candidates = c4[1];
        counts->n_keep = c4[2];
        counts->flags = c4[3];
    }
    const size_t n = c4[1], k = c4[2];
#define CP(dst, src, cnt, T) \
    if (dst && (cnt)) ADAS_HIP_TRY(hipMemcpy(dst, src, (cnt) * sizeof(T), hipMemcpyDeviceToHost))
    CP(cand_anchor, d.cand_anchor + b * cap, n, int);
    CP(cand_xywh, d.cand_xywh + b * cap * 4, n * 4, double);
    CP(cand_conf, d.cand_conf + b * cap, n, double);
    CP(cand_cls, d.cand_cls + b * cap, n, int);
    CP(keep, d.keep + b * cap, k, int);
    CP(det_xywh, d.det_xywh + b * cap * 4, k * 4, double);
    CP(det_conf, d.det_conf + b * cap, k, double);
    CP(det_cls, d.det_cls + b * cap, k, int);
    CP(det_xyxy_int, d.det_xyxy_i + b * cap * 4, k * 4, int);
#undef CP
    if (c4[3] & 1) {
        set_error("yolo_post: %d anchors over threshold exceed max_candidates=%d (frame %d)", c4[0], (int)cap, frame);
        return ADAS_ERR_CAPACITY;
    }
    return ADAS_OK;
}

int adas_yolo_post_fetch_dets(adas_yolo_post* h, int frame, adas_yolo_counts* counts, int32_t* keep, double* det_xywh, double* det_conf,
                              int32_t* det_cls, int32_t* det_xyxy_int) {
    ADAS_REQUIRE(h && frame >= 0 && frame < h->max_batch, ADAS_ERR_INVALID, "adas_yolo_post_fetch_dets: bad frame index");
    const size_t cap = h->p.max_candidates;
    yolo_pack_kernel<<<1, 256, 0, h->last>>>(h->dev, frame, h->msg);      // behind the NMS on its stream
    ADAS_HIP_TRY(hipGetLastError());
    const size_t first = cap < YOLO_MSG_FIRST ? cap : YOLO_MSG_FIRST;
    ADAS_HIP_TRY(hipMemcpyAsync(h->h_msg, h->msg, YOLO_MSG_HDR + first * YOLO_MSG_REC, hipMemcpyDeviceToHost, h->last));
    ADAS_HIP_TRY(hipStreamSynchronize(h->last));
    int c4[4];
    memcpy(c4, h->h_msg, 16);
    const size_t k = (size_t)c4[2];
    ADAS_REQUIRE(k <= cap, ADAS_ERR_INVALID, "adas_yolo_post_fetch_dets: corrupt survivor count %d", c4[2]);
    if (k > first)
        ADAS_HIP_TRY(hipMemcpy(h->h_msg + YOLO_MSG_HDR + first * YOLO_MSG_REC, h->msg + YOLO_MSG_HDR + first * YOLO_MSG_REC, (k - first) * YOLO_MSG_REC,
                               hipMemcpyDeviceToHost));
    if (counts) {
        counts->n_found = c4[0];
        counts->n_candidates = c4[1];
        counts->n_keep = c4[2];
        counts->flags = c4[3];
    }
    for (size_t i = 0; i < k; ++i) {
        const unsigned char* r = h->h_msg + YOLO_MSG_HDR + i * YOLO_MSG_REC;
        int w[6];
        memcpy(w, r + 40, 24);
        if (det_xywh) memcpy(det_xywh + i * 4, r, 32);
        if (det_conf) memcpy(det_conf + i, r + 32, 8);
        if (det_cls) det_cls[i] = w[0];
        if (keep) keep[i] = w[1];
        if (det_xyxy_int) memcpy(det_xyxy_int + i * 4, w + 2, 16);
    }
    if (c4[3] & 1) {
        set_error("yolo_post: %d anchors over threshold exceed max_candidates=%d (frame %d)", c4[0], (int)cap, frame);
        return ADAS_ERR_CAPACITY;
    }
    return ADAS_OK;
}

int adas_yolo_post_device_views(adas_yolo_post* h, const double** d_xyxy, const double** d_score, const int32_t** d_cls,
                                const int32_t** d_counts) {
    ADAS_REQUIRE(h, ADAS_ERR_INVALID, "null handle");
    if (d_xyxy) *d_xyxy = h->dev.det_xyxy_d;
    if (d_score) *d_score = h->dev.det_conf;
    if (d_cls) *d_cls = h->dev.det_cls;
    if (d_counts) *d_counts = h->dev.counts;
    return ADAS_OK;
}

int adas_yolo_post_capacity(const adas_yolo_post* h, int* max_candidates) {
    ADAS_REQUIRE(h && max_candidates, ADAS_ERR_INVALID, "null argument");
    *max_candidates = h->p.max_candidates;
    return ADAS_OK;
}

// ------------------------------------------------------------------------------- UFLD
int adas_effdet_post_create(const adas_effdet_post_params* p, int max_batch, adas_effdet_post** out) {
    ADAS_REQUIRE(p && out && max_batch > 0, ADAS_ERR_INVALID, "adas_effdet_post_create: bad argument");
    ADAS_REQUIRE(p->max_boxes >= 1 && p->max_boxes <= 4096, ADAS_ERR_INVALID, "max_boxes must be in [1, 4096]");
    ADAS_REQUIRE(p->ratio_h > 0 && p->ratio_w > 0, ADAS_ERR_INVALID, "scale ratios must be positive (fill them with adas_letterbox_params)");
    ADAS_REQUIRE(adas_device_count() > 0, ADAS_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    adas_effdet_post* h = new (std::nothrow) adas_effdet_post();
    ADAS_REQUIRE(h, ADAS_ERR_INVALID, "out of host memory");
    h->p = *p;
    h->max_batch = max_batch;
    h->last = 0;
    const size_t B = max_batch, cap = p->max_boxes;
    const size_t bytes = B * (cap * (16 + 4 + 4 + 16) + 8) + 8 * 256;
    if (hipMalloc(&h->arena, bytes) != hipSuccess) {
        delete h;
        return hip_fail(hipGetLastError(), "hipMalloc(effdet arena)", __FILE__, __LINE__);
    }
    hipMemset(h->arena, 0, bytes);
    unsigned char* q = (unsigned char*)h->arena;
    EffdetDev& d = h->dev;
    d.cfg = EffdetCfg{p->pad_h, p->pad_w, (float)p->ratio_h, (float)p->ratio_w, p->box_score, (int)cap};
    d.count = carve<int>(q, B);
    h->d_counts_in = carve<int>(q, B);
    d.counts_in = h->d_counts_in;
    d.xywh = carve<float>(q, B * cap * 4);
    d.conf = carve<float>(q, B * cap);
    d.cls = carve<int>(q, B * cap);
    d.xyxy_i = carve<int>(q, B * cap * 4);
    *out = h;
    return ADAS_OK;
}
int adas_effdet_post_destroy(adas_effdet_post* h) {
    if (!h) return ADAS_OK;
    hipFree(h->arena);
    delete h;
    return ADAS_OK;
}
int adas_effdet_post_run(adas_effdet_post* h, const float* d_boxes, const int32_t* d_ids, const float* d_confs, const int32_t* h_counts,
                         int batch, void* stream) {
    ADAS_REQUIRE(h && d_boxes && d_ids && d_confs && h_counts && batch > 0 && batch <= h->max_batch, ADAS_ERR_INVALID,
                 "adas_effdet_post_run: bad argument");
    for (int b = 0; b < batch; ++b)
        ADAS_REQUIRE(h_counts[b] >= 0 && h_counts[b] <= h->p.max_boxes, ADAS_ERR_CAPACITY, "frame %d carries %d detections, max_boxes is %d", b,
                     h_counts[b], h->p.max_boxes);
    hipStream_t st = (hipStream_t)stream;
    h->last = st;
    ADAS_HIP_TRY(hipMemcpyAsync(h->d_counts_in, h_counts, (size_t)batch * 4, hipMemcpyHostToDevice, st));
    EffdetDev d = h->dev;
    d.boxes = d_boxes; d.ids = d_ids; d.confs = d_confs;
    hipLaunchKernelGGL(effdet_post_kernel, dim3(batch), dim3(256), ((size_t)h->p.max_boxes + 1) * 4, st, d);
    ADAS_HIP_TRY(hipGetLastError());
    return ADAS_OK;
}
int adas_effdet_post_fetch(adas_effdet_post* h, int frame, int32_t* n_keep, float* xywh, float* conf, int32_t* class_id, int32_t* xyxy_int) {
    ADAS_REQUIRE(h && n_keep && frame >= 0 && frame < h->max_batch, ADAS_ERR_INVALID, "adas_effdet_post_fetch: bad argument");
    ADAS_HIP_TRY(hipStreamSynchronize(h->last));
    const EffdetDev& d = h->dev;
    const size_t cap = h->p.max_boxes, b = frame;
    int k = 0;
    ADAS_HIP_TRY(hipMemcpy(&k, d.count + b, 4, hipMemcpyDeviceToHost));
    *n_keep = k;
    if (k > 0) {
        if (xywh) ADAS_HIP_TRY(hipMemcpy(xywh, d.xywh + b * cap * 4, (size_t)k * 16, hipMemcpyDeviceToHost));
        if (conf) ADAS_HIP_TRY(hipMemcpy(conf, d.conf + b * cap, (size_t)k * 4, hipMemcpyDeviceToHost));
        if (class_id) ADAS_HIP_TRY(hipMemcpy(class_id, d.cls + b * cap, (size_t)k * 4, hipMemcpyDeviceToHost));
        if (xyxy_int) ADAS_HIP_TRY(hipMemcpy(xyxy_int, d.xyxy_i + b * cap * 4, (size_t)k * 16, hipMemcpyDeviceToHost));
    }
    return ADAS_OK;
}

int adas_effdet_tail_create(const adas_effdet_tail_params* p, int max_batch, adas_effdet_tail** out) {
    ADAS_REQUIRE(p && out && max_batch > 0, ADAS_ERR_INVALID, "adas_effdet_tail_create: bad argument");
    ADAS_REQUIRE(p->in_h >= 128 && p->in_w >= 128 && p->in_h % 128 == 0 && p->in_w % 128 == 0, ADAS_ERR_INVALID,
                 "the input size must be a multiple of 128 (five pyramid levels)");
    ADAS_REQUIRE(p->num_classes >= 1 && p->num_classes <= 4096, ADAS_ERR_INVALID, "num_classes must be in [1, 4096]");
    ADAS_REQUIRE(p->max_candidates >= 1 && p->max_candidates <= 3072 && p->max_det >= 1 && p->max_det <= p->max_candidates, ADAS_ERR_INVALID,
                 "max_candidates must be in [1, 3072] and max_det in [1, max_candidates]");
    ADAS_REQUIRE(p->score_thr >= 0 && p->score_thr < 1 && p->iou_thr > 0 && p->iou_thr <= 1 && p->anchor_scale > 0, ADAS_ERR_INVALID, "bad thresholds");
    ADAS_REQUIRE(adas_device_count() > 0, ADAS_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    adas_effdet_tail* h = new (std::nothrow) adas_effdet_tail();
    ADAS_REQUIRE(h, ADAS_ERR_INVALID, "out of host memory");
    h->p = *p;
    h->max_batch = max_batch;
    h->last = 0;
    const size_t B = max_batch, md = p->max_det;
    const size_t bytes = B * (md * (16 + 4 + 4) + 8) + 8 * 256;
    if (hipMalloc(&h->arena, bytes) != hipSuccess) {
        delete h;
        return hip_fail(hipGetLastError(), "hipMalloc(effdet tail arena)", __FILE__, __LINE__);
    }
    hipMemset(h->arena, 0, bytes);
    unsigned char* q = (unsigned char*)h->arena;
    EffdetTailDev& d = h->dev;
    d.cfg = EffdetTailCfg{p->in_h, p->in_w, p->num_classes, p->max_candidates, p->max_det, p->score_thr, p->iou_thr, p->anchor_scale};
    for (int l = 0; l < 5; ++l) d.rows[l] = (size_t)(p->in_h >> (3 + l)) * (size_t)(p->in_w >> (3 + l)) * 9;
    d.count = carve<int>(q, B * 2);
    d.boxes = carve<float>(q, B * md * 4);
    d.ids = carve<int>(q, B * md);
    d.confs = carve<float>(q, B * md);
    if (hipFuncSetAttribute((const void*)effdet_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        (void)hipFree(h->arena);
        delete h;
        return hip_fail(hipGetLastError(), "hipFuncSetAttribute(effdet_tail_kernel)", __FILE__, __LINE__);   // fails here, not at the first launch
    }
    *out = h;
    return ADAS_OK;
}
int adas_effdet_tail_destroy(adas_effdet_tail* h) {
    if (!h) return ADAS_OK;
    hipFree(h->arena);
    delete h;
    return ADAS_OK;
}
int adas_effdet_tail_run(adas_effdet_tail* h, const float* const* d_reg, const float* const* d_cls, int batch, void* stream) {
    ADAS_REQUIRE(h && d_reg && d_cls && batch > 0 && batch <= h->max_batch, ADAS_ERR_INVALID, "adas_effdet_tail_run: bad argument");
    EffdetTailDev d = h->dev;
    for (int l = 0; l < 5; ++l) {
        ADAS_REQUIRE(d_reg[l] && d_cls[l], ADAS_ERR_INVALID, "adas_effdet_tail_run: level %d tensor is null", l);
        d.reg[l] = d_reg[l]; d.cls[l] = d_cls[l];
    }
    hipStream_t st = (hipStream_t)stream;
    h->last = st;
    const size_t lds = effdet_tail_lds_bytes(h->p.max_candidates, 1024);
    hipLaunchKernelGGL(effdet_tail_kernel, dim3(batch), dim3(1024), lds, st, d);
    ADAS_HIP_TRY(hipGetLastError());
    return ADAS_OK;
}
int adas_effdet_tail_fetch(adas_effdet_tail* h, int frame, int32_t* n_det, float* boxes_xyxy, int32_t* class_id, float* conf, int32_t* n_candidates) {
    ADAS_REQUIRE(h && n_det && frame >= 0 && frame < h->max_batch, ADAS_ERR_INVALID, "adas_effdet_tail_fetch: bad argument");
    ADAS_HIP_TRY(hipStreamSynchronize(h->last));
    const EffdetTailDev& d = h->dev;
    const size_t md = h->p.max_det, b = frame;
    int cnt[2] = {0, 0};
    ADAS_HIP_TRY(hipMemcpy(cnt, d.count + b * 2, 8, hipMemcpyDeviceToHost));
    if (n_candidates) *n_candidates = cnt[1];
    ADAS_REQUIRE(cnt[1] <= h->p.max_candidates, ADAS_ERR_CAPACITY, "frame %d: %d anchors over score_thr, max_candidates is %d", frame, cnt[1],
                 h->p.max_candidates);
    *n_det = cnt[0];
    if (cnt[0] > 0) {
        if (boxes_xyxy) ADAS_HIP_TRY(hipMemcpy(boxes_xyxy, d.boxes + b * md * 4, (size_t)cnt[0] * 16, hipMemcpyDeviceToHost));
        if (class_id) ADAS_HIP_TRY(hipMemcpy(class_id, d.ids + b * md, (size_t)cnt[0] * 4, hipMemcpyDeviceToHost));
        if (conf) ADAS_HIP_TRY(hipMemcpy(conf, d.confs + b * md, (size_t)cnt[0] * 4, hipMemcpyDeviceToHost));
    }
    return ADAS_OK;
}
int adas_effdet_tail_device_views(adas_effdet_tail* h, const float** d_boxes, const int32_t** d_ids, const float** d_confs, const int32_t** d_counts) {
    ADAS_REQUIRE(h, ADAS_ERR_INVALID, "adas_effdet_tail_device_views: bad argument");
    if (d_boxes) *d_boxes = h->dev.boxes;
    if (d_ids) *d_ids = h->dev.ids;
    if (d_confs) *d_confs = h->dev.confs;
    if (d_counts) *d_counts = h->dev.count;
    return ADAS_OK;
}

int adas_ufld_decode_create(const adas_ufld_params* p, int max_batch, adas_ufld_decode** out) {
    ADAS_REQUIRE(p && out && max_batch > 0 && p->h_row_anchor && p->h_col_anchor, ADAS_ERR_INVALID, "adas_ufld_decode_create: bad argument");
    ADAS_REQUIRE(p->cls_row > 0 && p->cls_col > 0 && p->cls_row <= ADAS_UFLD_MAXPTS && p->cls_col <= ADAS_UFLD_MAXPTS,
                 ADAS_ERR_INVALID, "anchor counts must be in [1, %d]", ADAS_UFLD_MAXPTS);
    ADAS_REQUIRE(p->grid_row > 1 && p->grid_col > 1 && p->local_width >= 0 && p->local_width <= 3, ADAS_ERR_INVALID, "bad grid / local_width");
    ADAS_REQUIRE(p->num_lanes == 0 || (p->num_lanes >= 4 && p->num_lanes <= 16), ADAS_ERR_INVALID, "num_lanes must be 0 (= 4) or in [4, 16]");
    ADAS_REQUIRE(adas_device_count() > 0, ADAS_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    adas_ufld_decode* h = new (std::nothrow) adas_ufld_decode();
    ADAS_REQUIRE(h, ADAS_ERR_INVALID, "out of host memory");
    h->p = *p;
    if (h->p.num_lanes == 0) h->p.num_lanes = 4;
    h->max_batch = max_batch;
    h->v1 = 0;
    h->last = 0;
    const size_t B = max_batch;
    size_t bytes = (size_t)(p->cls_row + p->cls_col) * 8 + B * (32 + 4 * ADAS_UFLD_MAXPTS * 2 * 4) + 8 * 256;
    if (hipMalloc(&h->arena, bytes) != hipSuccess) {
        delete h;
        return hip_fail(hipGetLastError(), "hipMalloc(ufld arena)", __FILE__, __LINE__);
    }
    hipMemset(h->arena, 0, bytes);
    if (hipMalloc((void**)&h->msg, UFLD_MSG_INTS * 4) != hipSuccess || hipHostMalloc((void**)&h->h_msg, UFLD_MSG_INTS * 4, hipHostMallocDefault) != hipSuccess) {
        if (h->msg) hipFree(h->msg);
        hipFree(h->arena);
        delete h;
        return hip_fail(hipGetLastError(), "hipMalloc(ufld fetch message)", __FILE__, __LINE__);
    }
    unsigned char* q = (unsigned char*)h->arena;
    double* ra = carve<double>(q, p->cls_row);
    double* ca = carve<double>(q, p->cls_col);
    hipMemcpy(ra, p->h_row_anchor, p->cls_row * 8, hipMemcpyHostToDevice);
    hipMemcpy(ca, p->h_col_anchor, p->cls_col * 8, hipMemcpyHostToDevice);
    UfldDev& d = h->dev;
    d.cfg = UfldCfg{p->grid_row, p->cls_row, p->grid_col, p->cls_col, h->p.num_lanes, p->img_w, p->img_h, p->local_width, ra, ca};
    d.lane_cnt = carve<int>(q, B * 4);
    d.lane_det = carve<int>(q, B * 4);
    d.lane_pts = carve<int>(q, B * 4 * ADAS_UFLD_MAXPTS * 2);
    h->p.h_row_anchor = nullptr;
    h->p.h_col_anchor = nullptr;
    *out = h;
    return ADAS_OK;
}
int adas_ufld_decode_destroy(adas_ufld_decode* h) {
    if (!h) return ADAS_OK;
    if (h->msg) hipFree(h->msg);
    if (h->h_msg) hipHostFree(h->h_msg);
    hipFree(h->arena);
    delete h;
    return ADAS_OK;
}
int adas_ufld1_decode_create(const adas_ufld1_params* p, int max_batch, adas_ufld_decode** out) {
    ADAS_REQUIRE(p && out && max_batch > 0 && p->h_row_anchor, ADAS_ERR_INVALID, "adas_ufld1_decode_create: bad argument");
    ADAS_REQUIRE(p->griding_num > 2 && p->cls_num_per_lane > 0 && p->cls_num_per_lane <= ADAS_UFLD_MAXPTS, ADAS_ERR_INVALID,
                 "griding_num must be > 2 and cls_num_per_lane in [1, %d]", ADAS_UFLD_MAXPTS);
    ADAS_REQUIRE(p->cfg_img_w > 0 && p->cfg_img_h > 0 && p->input_w > 1 && p->input_h > 0 && p->src_w > 0 && p->src_h > 0, ADAS_ERR_INVALID,
                 "bad image geometry");
    ADAS_REQUIRE(adas_device_count() > 0, ADAS_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    adas_ufld_decode* h = new (std::nothrow) adas_ufld_decode();
    ADAS_REQUIRE(h, ADAS_ERR_INVALID, "out of host memory");
    memset(&h->p, 0, sizeof(h->p));
    h->max_batch = max_batch;
    h->v1 = 1;
    h->last = 0;
    const size_t B = max_batch;
    size_t bytes = (size_t)p->cls_num_per_lane * 8 + B * (32 + 4 * ADAS_UFLD_MAXPTS * 2 * 4) + 8 * 256;
    if (hipMalloc(&h->arena, bytes) != hipSuccess) {
        delete h;
        return hip_fail(hipGetLastError(), "hipMalloc(ufld1 arena)", __FILE__, __LINE__);
    }
    hipMemset(h->arena, 0, bytes);
    if (hipMalloc((void**)&h->msg, UFLD_MSG_INTS * 4) != hipSuccess || hipHostMalloc((void**)&h->h_msg, UFLD_MSG_INTS * 4, hipHostMallocDefault) != hipSuccess) {
        if (h->msg) hipFree(h->msg);
        hipFree(h->arena);
        delete h;
        return hip_fail(hipGetLastError(), "hipMalloc(ufld fetch message)", __FILE__, __LINE__);
    }
    unsigned char* q = (unsigned char*)h->arena;
    double* ra = carve<double>(q, p->cls_num_per_lane);
    hipMemcpy(ra, p->h_row_anchor, p->cls_num_per_lane * 8, hipMemcpyHostToDevice);
    Ufld1Dev& d = h->dev1;
    d.cfg = Ufld1Cfg{p->griding_num, p->cls_num_per_lane, 4, p->cfg_img_w, p->cfg_img_h, p->input_w, p->input_h, p->src_w, p->src_h, ra};
    d.lane_cnt = carve<int>(q, B * 4);
    d.lane_det = carve<int>(q, B * 4);
    d.lane_pts = carve<int>(q, B * 4 * ADAS_UFLD_MAXPTS * 2);
    h->dev.lane_cnt = d.lane_cnt;  // shared fetch path
    h->dev.lane_det = d.lane_det;
    h->dev.lane_pts = d.lane_pts;
    *out = h;
    return ADAS_OK;
}
int adas_ufld_decode_kind(const adas_ufld_decode* h) { return h ? (h->v1 ? 1 : 2) : 0; }
int adas_ufld_decode_expected_outputs(const adas_ufld_decode* h, int64_t dims[4][4]) {
    ADAS_REQUIRE(h && dims, ADAS_ERR_INVALID, "adas_ufld_decode_expected_outputs: null argument");
    if (h->v1) {  // ultrafastLaneDetector.py:73-75: one (1, griding_num + 1, cls_num_per_lane, 4) tensor
        dims[0][0] = 1; dims[0][1] = h->dev1.cfg.G + 1; dims[0][2] = h->dev1.cfg.K; dims[0][3] = h->dev1.cfg.L;
        return 1;
    }
    // model_culane.py:56-59: loc_row (1,G_r,K_r,4), loc_col (1,G_c,K_c,4), exist_row (1,2,K_r,4), exist_col (1,2,K_c,4)
    const int64_t g[4] = {h->p.grid_row, h->p.grid_col, 2, 2}, k[4] = {h->p.cls_row, h->p.cls_col, h->p.cls_row, h->p.cls_col};
    for (int i = 0; i < 4; ++i) { dims[i][0] = 1; dims[i][1] = g[i]; dims[i][2] = k[i]; dims[i][3] = h->p.num_lanes; }
    return 4;
}
int adas_ufld1_decode_set_source_size(adas_ufld_decode* h, int src_w, int src_h) {
    ADAS_REQUIRE(h && h->v1 && src_w > 0 && src_h > 0, ADAS_ERR_INVALID, "adas_ufld1_decode_set_source_size: bad argument");
    h->dev1.cfg.src_w = src_w;
    h->dev1.cfg.src_h = src_h;
    adas::bump_config_generation();
    return ADAS_OK;
}
int adas_ufld1_decode_run(adas_ufld_decode* h, const float* d_out, size_t batch_stride, int batch, void* stream) {
    ADAS_REQUIRE(h && h->v1 && d_out && batch > 0 && batch <= h->max_batch, ADAS_ERR_INVALID, "adas_ufld1_decode_run: bad argument");
    hipStream_t st = (hipStream_t)stream;
    h->last = st;
    Ufld1Dev d = h->dev1;
    d.out = d_out;
    d.stride = batch_stride;
    size_t lds = (size_t)d.cfg.K * d.cfg.L * 8 + 64;
    hipLaunchKernelGGL(ufld1_decode_kernel, dim3(batch), dim3(256), lds, st, d);
    ADAS_HIP_TRY(hipGetLastError());
    return ADAS_OK;
}
int adas_ufld_decode_run(adas_ufld_decode* h, const float* lr, const float* lc, const float* er, const float* ec,
                         size_t s_lr, size_t s_lc, size_t s_er, size_t s_ec, int batch, void* stream) {
    ADAS_REQUIRE(h && !h->v1 && lr && lc && er && ec && batch > 0 && batch <= h->max_batch, ADAS_ERR_INVALID, "adas_ufld_decode_run: bad argument");
    hipStream_t st = (hipStream_t)stream;
    h->last = st;
    UfldDev d = h->dev;
    d.loc_row = lr; d.loc_col = lc; d.exist_row = er; d.exist_col = ec;
    d.s_lr = s_lr; d.s_lc = s_lc; d.s_er = s_er; d.s_ec = s_ec;
    size_t lds = UfldLds::bytes(h->p.cls_row, h->p.cls_col, h->p.num_lanes);
    hipLaunchKernelGGL(ufld_decode_kernel, dim3(batch), dim3(640), lds, st, d);
    ADAS_HIP_TRY(hipGetLastError());
    return ADAS_OK;
}
int adas_ufld_decode_fetch(adas_ufld_decode* h, int frame, int32_t* points, int32_t* counts, int32_t* detected) {
    ADAS_REQUIRE(h && frame >= 0 && frame < h->max_batch, ADAS_ERR_INVALID, "adas_ufld_decode_fetch: bad frame index");
    const UfldDev& d = h->dev;
    // one pack kernel behind the decode on its stream + ONE copy (was: three synchronous copies)
    ufld_pack_kernel<<<1, 256, 0, h->last>>>(d.lane_cnt, d.lane_det, d.lane_pts, frame, h->msg);
    ADAS_HIP_TRY(hipGetLastError());
    ADAS_HIP_TRY(hipMemcpyAsync(h->h_msg, h->msg, UFLD_MSG_INTS * 4, hipMemcpyDeviceToHost, h->last));
    ADAS_HIP_TRY(hipStreamSynchronize(h->last));
    if (counts) memcpy(counts, h->h_msg, 16);
    if (detected) memcpy(detected, h->h_msg + 4, 16);
    if (points) memcpy(points, h->h_msg + 8, 4 * ADAS_UFLD_MAXPTS * 2 * 4);
    return ADAS_OK;
}

int adas_ufld_decode_upload(adas_ufld_decode* h, int frame, const int32_t* points, const int32_t* counts, const int32_t* detected) {
    ADAS_REQUIRE(h && points && counts && detected && frame >= 0 && frame < h->max_batch, ADAS_ERR_INVALID, "adas_ufld_decode_upload: bad argument");
    for (int l = 0; l < 4; ++l)
        ADAS_REQUIRE(counts[l] >= 0 && counts[l] <= ADAS_UFLD_MAXPTS, ADAS_ERR_INVALID, "lane %d: %d points (max %d)", l, counts[l], ADAS_UFLD_MAXPTS);
    ADAS_HIP_TRY(hipStreamSynchronize(h->last));
    const UfldDev& d = h->dev;
    ADAS_HIP_TRY(hipMemcpy(d.lane_pts + (size_t)frame * 4 * ADAS_UFLD_MAXPTS * 2, points, 4 * ADAS_UFLD_MAXPTS * 2 * 4, hipMemcpyHostToDevice));
    ADAS_HIP_TRY(hipMemcpy(d.lane_cnt + frame * 4, counts, 16, hipMemcpyHostToDevice));
    ADAS_HIP_TRY(hipMemcpy(d.lane_det + frame * 4, detected, 16, hipMemcpyHostToDevice));
    return ADAS_OK;
}

// ------------------------------------------------------------------------------- lane geometry
int adas_lane_geometry_create(const adas_lane_geometry_params* p, int max_batch, adas_lane_geometry** out) {
    ADAS_REQUIRE(p && out && max_batch > 0, ADAS_ERR_INVALID, "adas_lane_geometry_create: bad argument");
    ADAS_REQUIRE(p->img_h >= 8 && p->img_h <= 4320 && p->bird_h >= 8 && p->bird_h <= 4320 && p->bird_w > 0, ADAS_ERR_INVALID,
                 "image heights must be in [8, 4320]");
    ADAS_REQUIRE(adas_device_count() > 0, ADAS_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    const size_t lds = lane_lds_bytes(p->img_h, p->bird_h);
    ADAS_REQUIRE(lds <= 150 * 1024, ADAS_ERR_CAPACITY, "image height %d needs %zu bytes of LDS", p->img_h > p->bird_h ? p->img_h : p->bird_h, lds);
    adas_lane_geometry* h = new (std::nothrow) adas_lane_geometry();
    ADAS_REQUIRE(h, ADAS_ERR_INVALID, "out of host memory");
    h->max_batch = max_batch;
    h->last = 0;
    const size_t B = max_batch, H = p->img_h;
    size_t bytes = B * (8 * 4 + 2 * 8 + 4 * H * 4 + 4 * ADAS_LANE_MAXPTS * 2 * 4 + 2 * H * 8 + 2 * H * 4) + 16 * 256;
    if (hipMalloc(&h->arena, bytes) != hipSuccess) {
        delete h;
        return hip_fail(hipGetLastError(), "hipMalloc(lane geometry arena)", __FILE__, __LINE__);
    }
    hipMemset(h->arena, 0, bytes);
    unsigned char* q = (unsigned char*)h->arena;
    LaneGeomDev& d = h->dev;
    d.cfg.img_h = p->img_h; d.cfg.bird_w = p->bird_w; d.cfg.bird_h = p->bird_h; d.cfg.adjust = p->adjust_lanes ? 1 : 0;
    for (int i = 0; i < 9; ++i) d.cfg.M[i] = p->M[i];
    d.vals = carve<double>(q, B * 2);
    d.fx = carve<double>(q, B * 2 * H);
    d.hdr = carve<int>(q, B * 8);
    d.area = carve<int>(q, B * 4 * H);
    d.bird = carve<int>(q, B * 4 * ADAS_LANE_MAXPTS * 2);
    d.idx = carve<int>(q, B * 2 * H);
    if (hipFuncSetAttribute((const void*)lane_geometry_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
        hipFree(h->arena);
        delete h;
        return hip_fail(hipGetLastError(), "hipFuncSetAttribute(lane_geometry_kernel)", __FILE__, __LINE__);
    }
    *out = h;
    return ADAS_OK;
}
int adas_lane_geometry_destroy(adas_lane_geometry* h) {
    if (!h) return ADAS_OK;
    if (h->msg) hipFree(h->msg);
    if (h->h_msg) hipHostFree(h->h_msg);
    hipFree(h->arena);
    delete h;
    return ADAS_OK;
}
int adas_lane_geometry_set_matrix(adas_lane_geometry* h, const double* M9) {
    ADAS_REQUIRE(h && M9, ADAS_ERR_INVALID, "adas_lane_geometry_set_matrix: bad argument");
    for (int i = 0; i < 9; ++i) h->dev.cfg.M[i] = M9[i];
    adas::bump_config_generation();
    return ADAS_OK;
}
int adas_lane_geometry_run(adas_lane_geometry* h, const adas_ufld_decode* decode, int adjust_lanes, int batch, void* stream) {
    ADAS_REQUIRE(h && decode && batch > 0 && batch <= h->max_batch && batch <= decode->max_batch, ADAS_ERR_INVALID,
                 "adas_lane_geometry_run: bad argument");
    hipStream_t st = (hipStream_t)stream;
    h->last = st;
    LaneGeomDev d = h->dev;
    if (adjust_lanes >= 0) d.cfg.adjust = adjust_lanes ? 1 : 0;
    d.lane_cnt = decode->dev.lane_cnt;
    d.lane_det = decode->dev.lane_det;
    d.lane_pts = decode->dev.lane_pts;
    hipLaunchKernelGGL(lane_geometry_kernel, dim3(batch), dim3(256), lane_lds_bytes(d.cfg.img_h, d.cfg.bird_h), st, d);
    ADAS_HIP_TRY(hipGetLastError());
    return ADAS_OK;
}
int adas_lane_geometry_fetch(adas_lane_geometry* h, int frame, adas_lane_geometry_result* res, int32_t* area_points, int32_t* bird_points) {
    ADAS_REQUIRE(h && res && frame >= 0 && frame < h->max_batch, ADAS_ERR_INVALID, "adas_lane_geometry_fetch: bad argument");
    const LaneGeomDev& d = h->dev;
    if (!h->msg) ADAS_HIP_TRY(hipMalloc((void**)&h->msg, LGEO_MSG_INTS * 4));          // first fetch: the message buffers
    if (!h->h_msg) ADAS_HIP_TRY(hipHostMalloc((void**)&h->h_msg, LGEO_MSG_INTS * 4, hipHostMallocDefault));
    lane_geometry_pack_kernel<<<1, 256, 0, h->last>>>(d.hdr, d.vals, d.bird, frame, h->msg);   // behind the geometry kernel on its stream
    ADAS_HIP_TRY(hipGetLastError());
    ADAS_HIP_TRY(hipMemcpyAsync(h->h_msg, h->msg, LGEO_MSG_INTS * 4, hipMemcpyDeviceToHost, h->last));
    ADAS_HIP_TRY(hipStreamSynchronize(h->last));
    int hdr[8];
    double vals[2];
    memcpy(hdr, h->h_msg, sizeof(hdr));
    memcpy(vals, h->h_msg + 8, sizeof(vals));
    res->area_status = hdr[0]; res->n_area_left = hdr[1]; res->n_area_right = hdr[2]; res->direction = hdr[3];
    for (int l = 0; l < 4; ++l) res->bird_counts[l] = hdr[4 + l];
    res->curvature = vals[0]; res->offset = vals[1];
    const size_t H = d.cfg.img_h;
    if (area_points && hdr[1] + hdr[2] > 0)
        ADAS_HIP_TRY(hipMemcpy(area_points, d.area + (size_t)frame * 4 * H, (size_t)(hdr[1] + hdr[2]) * 8, hipMemcpyDeviceToHost));
    if (bird_points) memcpy(bird_points, h->h_msg + 12, 4 * ADAS_LANE_MAXPTS * 2 * 4);
    return ADAS_OK;
}

// ------------------------------------------------------------------------------- ByteTrack
int adas_bytetrack_create(const adas_bytetrack_params* p, int n_streams, adas_bytetrack** out) {
    ADAS_REQUIRE(p && out && n_streams > 0, ADAS_ERR_INVALID, "adas_bytetrack_create: bad argument");
    ADAS_REQUIRE(p->max_tracks >= 8 && p->max_tracks <= 1024 && p->max_dets >= 8 && p->max_dets <= 2048, ADAS_ERR_INVALID,
                 "max_tracks must be in [8,1024], max_dets in [8,2048]");
    ADAS_REQUIRE(adas_device_count() > 0, ADAS_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    static_assert(sizeof(adas_track) == sizeof(BtOut), "adas_track layout");
    static_assert(sizeof(adas_track_header) == sizeof(BtHeader), "adas_track_header layout");
    adas_bytetrack* h = new (std::nothrow) adas_bytetrack();
    ADAS_REQUIRE(h, ADAS_ERR_INVALID, "out of host memory");
    h->p = *p;
    h->n_streams = n_streams;
    h->last = 0;
    BtDev& d = h->dev;
    d.P = BtParams{p->track_thresh, p->track_thresh + 0.1, p->match_thresh,
                   (int)(p->frame_rate / 30.0 * p->track_buffer), p->max_tracks, p->max_dets};  // byteTracker.py:48-50
    d.stream_bytes = bt_stream_bytes(p->max_tracks, p->max_dets);
    size_t stage = (size_t)p->max_dets * (32 + 8 + 4) + 256 * 4;
    stage = bt_align(stage) + bt_align((size_t)p->max_tracks * 4) + (size_t)p->max_tracks * ADAS_BT_TRAJ * 32;   // + the trajectory gather buffer
    size_t bytes = d.stream_bytes * n_streams + stage;
    if (hipMalloc(&h->arena, bytes) != hipSuccess) {
        delete h;
        return hip_fail(hipGetLastError(), "hipMalloc(bytetrack arena)", __FILE__, __LINE__);
    }
    hipMemset(h->arena, 0, bytes);
    d.base = (unsigned char*)h->arena;
    h->h_stage_d = (double*)(d.base + d.stream_bytes * n_streams);
    size_t lds = BtLds::bytes(p->max_tracks, p->max_dets, 256);
    if (hipFuncSetAttribute((const void*)bytetrack_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        hipFree(h->arena);
        delete h;
        return hip_fail(hipGetLastError(), "hipFuncSetAttribute(bytetrack_update_kernel)", __FILE__, __LINE__);
    }
    *out = h;
    return ADAS_OK;
}
int adas_bytetrack_destroy(adas_bytetrack* h) {
    if (!h) return ADAS_OK;
    if (h->snap) (void)hipFree(h->snap);
    hipFree(h->arena);
    delete h;
    return ADAS_OK;
}
int adas_bytetrack_reset(adas_bytetrack* h, int stream_index) {
    ADAS_REQUIRE(h && stream_index >= -1 && stream_index < h->n_streams, ADAS_ERR_INVALID, "adas_bytetrack_reset: bad stream index");
    BtDev d = h->dev;
    d.first_stream = stream_index < 0 ? 0 : stream_index;
    int n = stream_index < 0 ? h->n_streams : 1;
    hipLaunchKernelGGL(bytetrack_reset_kernel, dim3(n), dim3(256), 0, h->last, d);
    ADAS_HIP_TRY(hipGetLastError());
    return ADAS_OK;
}
int adas_bytetrack_update_host(adas_bytetrack* h, int stream_index, const double* xyxy, const double* scores,
                               const int32_t* cls, int n) {
    ADAS_REQUIRE(h && stream_index >= 0 && stream_index < h->n_streams && n >= 0, ADAS_ERR_INVALID, "adas_bytetrack_update_host: bad argument");
    ADAS_REQUIRE(n == 0 || (xyxy && scores && cls), ADAS_ERR_INVALID, "null detection arrays");
    ADAS_REQUIRE(n <= h->p.max_dets, ADAS_ERR_CAPACITY, "%d detections exceed max_dets=%d", n, h->p.max_dets);
    const int MD = h->p.max_dets;
    double* dx = h->h_stage_d;
    double* ds = dx + (size_t)MD * 4;
    int* dc = (int*)(ds + MD);
    int* dn = dc + MD;
    hipStream_t st = h->last;
    if (n) {
        ADAS_HIP_TRY(hipMemcpyAsync(dx, xyxy, (size_t)n * 32, hipMemcpyHostToDevice, st));
        ADAS_HIP_TRY(hipMemcpyAsync(ds, scores, (size_t)n * 8, hipMemcpyHostToDevice, st));
        ADAS_HIP_TRY(hipMemcpyAsync(dc, cls, (size_t)n * 4, hipMemcpyHostToDevice, st));
    }
    ADAS_HIP_TRY(hipMemcpyAsync(dn, &n, 4, hipMemcpyHostToDevice, st));
    ADAS_HIP_TRY(hipStreamSynchronize(st));  // n lives on the caller's stack
    BtDev d = h->dev;
    d.xyxy = dx; d.score = ds; d.cls = dc; d.counts = dn;
    d.det_stride = MD; d.count_stride = 1; d.count_index = 0; d.first_stream = stream_index;
    size_t lds = BtLds::bytes(h->p.max_tracks, h->p.max_dets, 256);
    hipLaunchKernelGGL(bytetrack_update_kernel, dim3(1), dim3(256), lds, st, d);
    ADAS_HIP_TRY(hipGetLastError());
    return ADAS_OK;
}
int adas_bytetrack_update_device(adas_bytetrack* h, const double* d_xyxy, const double* d_scores, const int32_t* d_cls,
                                 const int32_t* d_counts, int det_stride, int count_stride, int count_index, int n_streams,
                                 void* stream) {
    ADAS_REQUIRE(h && d_xyxy && d_scores && d_cls && d_counts && n_streams > 0 && n_streams <= h->n_streams, ADAS_ERR_INVALID,
                 "adas_bytetrack_update_device: bad argument");
    hipStream_t st = (hipStream_t)stream;
    h->last = st;
    BtDev d = h->dev;
    d.xyxy = d_xyxy; d.score = d_scores; d.cls = d_cls; d.counts = d_counts;
    d.det_stride = det_stride; d.count_stride = count_stride; d.count_index = count_index; d.first_stream = 0;
    size_t lds = BtLds::bytes(h->p.max_tracks, h->p.max_dets, 256);
    hipLaunchKernelGGL(bytetrack_update_kernel, dim3(n_streams), dim3(256), lds, st, d);
    ADAS_HIP_TRY(hipGetLastError());
    return ADAS_OK;
}
int adas_bytetrack_reserve_frames(adas_bytetrack* h, int n_frames, int n_streams) {
    ADAS_REQUIRE(h && n_frames >= 1 && n_streams >= 1 && n_streams <= h->n_streams, ADAS_ERR_INVALID, "adas_bytetrack_reserve_frames: bad argument");
    if (n_frames <= h->snap_frames && n_streams == h->snap_streams) return ADAS_OK;
    ADAS_HIP_TRY(hipStreamSynchronize(h->last));
    if (h->snap) (void)hipFree(h->snap);
    h->snap = nullptr; h->snap_frames = h->snap_streams = 0;
    ADAS_HIP_TRY(hipMalloc(&h->snap, bt_snap_bytes(h->p.max_tracks) * (size_t)n_frames * (size_t)n_streams));
    ADAS_HIP_TRY(hipMemset(h->snap, 0, bt_snap_bytes(h->p.max_tracks) * (size_t)n_frames * (size_t)n_streams));
    h->snap_frames = n_frames; h->snap_streams = n_streams;
    adas::bump_config_generation();   // a step captured with the old store's address is re-captured (pipeline.cpp replay_step)
    return ADAS_OK;
}
int adas_bytetrack_update_device_frames(adas_bytetrack* h, const double* d_xyxy, const double* d_scores, const int32_t* d_cls,
                                        const int32_t* d_counts, int det_stride, int count_stride, int count_index, int n_streams, int n_frames,
                                        void* stream) {
    ADAS_REQUIRE(h && d_xyxy && d_scores && d_cls && d_counts && n_streams > 0 && n_streams <= h->n_streams && n_frames > 0, ADAS_ERR_INVALID,
                 "adas_bytetrack_update_device_frames: bad argument");
    hipStream_t st = (hipStream_t)stream;
    h->last = st;
    BtDev d = h->dev;
    d.xyxy = d_xyxy; d.score = d_scores; d.cls = d_cls; d.counts = d_counts;
    d.det_stride = det_stride; d.count_stride = count_stride; d.count_index = count_index; d.first_stream = 0;
    d.n_frames = n_frames; d.frame_slabs = n_streams;
    if (n_frames > 1) {   // keep every frame's message fetchable (adas_bytetrack_fetch_frame)
        if (n_frames > h->snap_frames || n_streams != h->snap_streams) {   // (re)allocation: never inside a stream capture (adas_pipeline_create reserves)
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(st, &cs);
            ADAS_REQUIRE(cs == hipStreamCaptureStatusNone, ADAS_ERR_INVALID,
                         "call adas_bytetrack_reserve_frames(%d frames, %d streams) before capturing a micro-batched step", n_frames, n_streams);
            int rc = adas_bytetrack_reserve_frames(h, n_frames, n_streams);
            if (rc) return rc;
        }
        d.snap = (unsigned char*)h->snap; d.snap_bytes = bt_snap_bytes(h->p.max_tracks);
    }
    h->last_frames = n_frames > 1 ? n_frames : 0;   // what adas_bytetrack_fetch_frame may ask for (older snapshots in the store are stale)
    size_t lds = BtLds::bytes(h->p.max_tracks, h->p.max_dets, 256);
    hipLaunchKernelGGL(bytetrack_update_kernel, dim3(n_streams), dim3(256), lds, st, d);
    ADAS_HIP_TRY(hipGetLastError());
    return ADAS_OK;
}
int adas_bytetrack_fetch_frame(adas_bytetrack* h, int stream_index, int frame, adas_track_header* hdr, adas_track* tracks, int max_tracks) {
    ADAS_REQUIRE(h && hdr && stream_index >= 0 && stream_index < h->n_streams, ADAS_ERR_INVALID, "adas_bytetrack_fetch_frame: bad argument");
    ADAS_REQUIRE(h->snap && frame >= 0 && frame < h->snap_frames && frame < h->last_frames && stream_index < h->snap_streams, ADAS_ERR_INVALID,
                 "adas_bytetrack_fetch_frame: frame %d of stream %d is not part of the last adas_bytetrack_update_device_frames launch (%d frames x %d streams)",
                 frame, stream_index, h->last_frames, h->snap_streams);
    ADAS_HIP_TRY(hipStreamSynchronize(h->last));
    const unsigned char* q = (const unsigned char*)h->snap + ((size_t)frame * h->snap_streams + stream_index) * bt_snap_bytes(h->p.max_tracks);
    ADAS_HIP_TRY(hipMemcpy(hdr, q, sizeof(BtHeader), hipMemcpyDeviceToHost));
    const int n = hdr->n_tracked + hdr->n_lost;
    if (tracks && n > 0) {
        ADAS_REQUIRE(n <= max_tracks, ADAS_ERR_CAPACITY, "fetch buffer holds %d tracks, need %d", max_tracks, n);
        ADAS_HIP_TRY(hipMemcpy(tracks, q + bt_align(sizeof(BtHeader)), (size_t)n * sizeof(BtOut), hipMemcpyDeviceToHost));
    }
    if (hdr->err & (BT_ERR_DET_OVERFLOW | BT_ERR_TRACK_OVERFLOW | BT_ERR_HIST_OVERFLOW)) {   // as adas_bytetrack_fetch
        set_error("bytetrack stream %d, frame %d: capacity exceeded (err bits 0x%x)", stream_index, frame, hdr->err);
        return ADAS_ERR_CAPACITY;
    }
    return ADAS_OK;
}
int adas_bytetrack_fetch_trajectories(adas_bytetrack* h, int stream_index, int32_t* lens, double* tlbr, int max_tracks, int32_t* n_tracks) {
    ADAS_REQUIRE(h && n_tracks && stream_index >= 0 && stream_index < h->n_streams && max_tracks >= 0, ADAS_ERR_INVALID,
                 "adas_bytetrack_fetch_trajectories: bad argument");
    static_assert(ADAS_TRAJECTORY_LEN == ADAS_BT_TRAJ, "trajectory depth");
    const size_t MT = h->p.max_tracks;
    size_t stage0 = bt_align((size_t)h->p.max_dets * (32 + 8 + 4) + 256 * 4);
    int* d_lens = (int*)((unsigned char*)h->h_stage_d + stage0);
    double* d_out = (double*)((unsigned char*)d_lens + bt_align(MT * 4));
    hipLaunchKernelGGL(bytetrack_traj_kernel, dim3(1), dim3(256), 0, h->last, h->dev, stream_index, d_lens, d_out);   // behind the last update
    ADAS_HIP_TRY(hipGetLastError());
    ADAS_HIP_TRY(hipStreamSynchronize(h->last));
    BtStream S = bt_view(h->dev.base + (size_t)stream_index * h->dev.stream_bytes, h->p.max_tracks, h->p.max_dets);
    BtHeader hdr;
    ADAS_HIP_TRY(hipMemcpy(&hdr, S.hdr, sizeof(BtHeader), hipMemcpyDeviceToHost));
    const int n = hdr.n_tracked + hdr.n_lost;
    *n_tracks = n;
    if (n > 0 && (lens || tlbr)) {
        ADAS_REQUIRE(n <= max_tracks, ADAS_ERR_CAPACITY, "fetch buffer holds %d tracks, need %d", max_tracks, n);
        if (lens) ADAS_HIP_TRY(hipMemcpy(lens, d_lens, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (tlbr) ADAS_HIP_TRY(hipMemcpy(tlbr, d_out, (size_t)n * ADAS_BT_TRAJ * 32, hipMemcpyDeviceToHost));
    }
    return ADAS_OK;
}

int adas_bytetrack_fetch(adas_bytetrack* h, int stream_index, adas_track_header* hdr, adas_track* tracks, int max_tracks) {
    ADAS_REQUIRE(h && hdr && stream_index >= 0 && stream_index < h->n_streams, ADAS_ERR_INVALID, "adas_bytetrack_fetch: bad argument");
    ADAS_HIP_TRY(hipStreamSynchronize(h->last));
    BtStream S = bt_view(h->dev.base + (size_t)stream_index * h->dev.stream_bytes, h->p.max_tracks, h->p.max_dets);
    ADAS_HIP_TRY(hipMemcpy(hdr, S.hdr, sizeof(BtHeader), hipMemcpyDeviceToHost));
    int n = hdr->n_tracked + hdr->n_lost;
    if (tracks && n > 0) {
        ADAS_REQUIRE(n <= max_tracks, ADAS_ERR_CAPACITY, "fetch buffer holds %d tracks, need %d", max_tracks, n);
        ADAS_HIP_TRY(hipMemcpy(tracks, S.out, (size_t)n * sizeof(BtOut), hipMemcpyDeviceToHost));
    }
    if (hdr->err & (BT_ERR_DET_OVERFLOW | BT_ERR_TRACK_OVERFLOW | BT_ERR_HIST_OVERFLOW)) {
        set_error("bytetrack stream %d: capacity exceeded (err bits 0x%x)", stream_index, hdr->err);
        return ADAS_ERR_CAPACITY;
    }
    return ADAS_OK;
}
}  // extern "C"
