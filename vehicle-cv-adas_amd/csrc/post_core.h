// post_core.h -- GPU-resident post-processing logic of the per-frame ADAS path.
//
// One workgroup per frame / per video stream executes these routines.  All
// decision arithmetic is IEEE fp64 with contraction OFF so that survivor
// indices and track ids are bit-exact with the reference's pinned NumPy
// environment (SURVEY.md finding 5, Appendix A3/A6/A7).
//
// The routines are written against a tiny block-execution context (tid,
// nthr, sync).  hipcc instantiates them inside __global__ kernels
// (post_kernels.hip).  tests/hostemu compiles the very same source with g++ and
// nthr == 1 to debug the *logic* on the GPU-less build container; that build is
// test scaffolding only and is never loaded by the product.
#pragma once
#include <stdint.h>
#include <math.h>
#include <float.h>

#if defined(__HIPCC__)
#define ADAS_DEV __device__ __forceinline__
#define ADAS_HD __host__ __device__ inline
#pragma clang fp contract(off)
#else
#define ADAS_DEV inline
#define ADAS_HD inline
#endif

namespace adas {

struct Ctx {
    int tid, nthr;
    ADAS_DEV void sync() const {
#if defined(__HIP_DEVICE_COMPILE__)
        __syncthreads();
#endif
    }
};

#define ADAS_PAR_FOR(c, i, lo, hi) for (int i = (lo) + (c).tid; i < (hi); i += (c).nthr)

// ---------------------------------------------------------------------------
// block-wide "first index of the maximum" over a[lo,hi)  (np.argmax semantics)
// red_v/red_i: LDS scratch of nthr entries.
// ---------------------------------------------------------------------------
ADAS_DEV void block_argmax_first(const Ctx& c, const double* a, int lo, int hi, double* red_v, int* red_i,
                                 double& out_v, int& out_i) {
    double bv = -DBL_MAX;
    int bi = 0x7fffffff;
    for (int j = lo + c.tid; j < hi; j += c.nthr) {
        double v = a[j];
        if (v > bv || bi == 0x7fffffff) {
            bv = v;
            bi = j;
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // wave64 shuffle reduce, then one LDS hop across waves
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_down(bv, off, 64);
        int oi = __shfl_down(bi, off, 64);
        if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
    }
    int lane = c.tid & 63, wv = c.tid >> 6, nw = (c.nthr + 63) >> 6;
    if (lane == 0) {
        red_v[wv] = bv;
        red_i[wv] = bi;
    }
    c.sync();
    if (c.tid == 0) {
        for (int w = 1; w < nw; ++w) {
            double ov = red_v[w];
            int oi = red_i[w];
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) {
                bv = ov;
                bi = oi;
            }
        }
        red_v[0] = bv;
        red_i[0] = bi;
    }
    c.sync();
    out_v = red_v[0];
    out_i = red_i[0];
    c.sync();
#else
    (void)red_v;
    (void)red_i;
    out_v = bv;
    out_i = bi;
#endif
}

// ---------------------------------------------------------------------------
// wave64 all-lanes reductions / broadcasts on VALU lane exchanges: DPP inside the 16-lane rows, v_permlane16_swap and
// v_permlane32_swap across them, v_readlane for uniform-source broadcasts.  About 30 VALU instructions per reduction
// instead of six dependent ds_bpermute round trips (~150 cycles each) -- these sit inside the sequential loops of the
// NMS and of the assignment solver.
// ---------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
ADAS_DEV unsigned long long wv_pack(int hi, int lo) { return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo; }
template <int CTRL>
ADAS_DEV unsigned long long wv_dpp64(unsigned long long v) {
    int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return wv_pack(hi, lo);
}
template <class Op>
ADAS_DEV unsigned long long wave_allreduce_u64(unsigned long long v, Op op) {
    v = op(v, wv_dpp64<0xB1>(v));   // quad_perm [1,0,3,2]: lane ^ 1
    v = op(v, wv_dpp64<0x4E>(v));   // quad_perm [2,3,0,1]: lane ^ 2
    v = op(v, wv_dpp64<0x141>(v));  // row_half_mirror: quads of each half row
    v = op(v, wv_dpp64<0x140>(v));  // row_mirror: the two halves of a row
    {
        const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = op(wv_pack((int)b[0], (int)a[0]), wv_pack((int)b[1], (int)a[1]));  // rows 0|1 and 2|3
    }
    {
        const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = op(wv_pack((int)b[0], (int)a[0]), wv_pack((int)b[1], (int)a[1]));  // the two half waves
    }
    return v;
}
ADAS_DEV unsigned long long wave_max_u64(unsigned long long v) {
    return wave_allreduce_u64(v, [](unsigned long long a, unsigned long long b) { return a > b ? a : b; });
}
ADAS_DEV double wave_min_f64(double v) {  // no NaNs expected; of two equal values either bit pattern may come back
    const unsigned long long r = wave_allreduce_u64((unsigned long long)__double_as_longlong(v), [](unsigned long long a, unsigned long long b) {
        return __longlong_as_double((long long)a) <= __longlong_as_double((long long)b) ? a : b;
    });
    return __longlong_as_double((long long)r);
}
// value of lane `src` (wave-uniform) in every lane
ADAS_DEV int wave_read_i32(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }
ADAS_DEV double wave_read_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int s = __builtin_amdgcn_readfirstlane(src);
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)b, s), hi = __builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), s);
    return __longlong_as_double((long long)wv_pack(hi, lo));
}
#endif

// ===========================================================================
// YOLO: ordered compaction -> box decode -> inverse letterbox -> NMS -> gather
// replaces yoloDetector.py:120-133 (threshold + box), utils.py:70-87,
// utils.py:161-256 / 105-159, yoloDetector.py:135-157 + core.py:18-23
// ===========================================================================
struct YoloPostCfg {
    int layout;  // 0: v8-family head [4+nc][A];  1: v5-family head [A][5+nc];  2: v5-lite (layout 1 + grid decode)
    int A, nc;
    double box_score, iou_thr;
    int nms_mode;  // 0: reference (fast_soft_nms as called in production); 1: greedy (fast_nms)
    int pad_h, pad_w;
    double ratio_h, ratio_w;
    int cap;  // candidate capacity per frame
    int in_h, in_w;  // network input size; only the v5-lite grid decode reads it
};

// YoloLiteParameters.lite_postprocess (yoloDetector.py:35-49) for one head row: three levels (stride 8/16/32) of
// na = 3 anchors x (h*w) cells, fp32 arithmetic in the reference's operation order.  The reference builds its grid as
// meshgrid(arange(h), arange(w)) flattened, i.e. cell n -> (n % h, n / h); that equals the usual (n % w, n / w) only
// for square inputs and is reproduced as is.  Rows past the three levels are left untouched, as there.
ADAS_DEV void yolo_lite_decode(int in_h, int in_w, int row, float& x, float& y, float& w, float& h) {
    const float anchors[3][6] = {{10, 13, 16, 30, 33, 23}, {30, 61, 62, 45, 59, 119}, {116, 90, 156, 198, 373, 326}};
    for (int i = 0; i < 3; ++i) {
        const int stride = 8 << i;
        const int gh = in_h / stride, gw = in_w / stride, cells = gh * gw;
        if (row < 3 * cells) {
            const int ai = row / cells, n = row % cells;
            const float gx = (float)(n % gh), gy = (float)(n / gh);
            const float fs = (float)stride;
            x = ((x * 2.f - 0.5f) + gx) * fs;
            y = ((y * 2.f - 0.5f) + gy) * fs;
            const float w2 = w * 2.f, h2 = h * 2.f;
            w = (w2 * w2) * anchors[i][2 * ai];
            h = (h2 * h2) * anchors[i][2 * ai + 1];
            return;
        }
        row -= 3 * cells;
    }
}

// counts[]: 0 = candidates found (uncapped), 1 = candidates stored, 2 = survivors, 3 = flags (bit0 overflow)
struct YoloPostFrame {
    const float* head;       // this frame's head tensor (reference layout)
    const float* best_conf;  // [A] from the scan kernel
    const int* best_cls;     // [A]
    int* counts;
    int* cand_anchor;
    double* cand_xywh;  // [cap][4]
    double* cand_conf;  // [cap]
    int* cand_cls;      // [cap]
    int* keep;          // [cap] indices into candidates (selection order; may repeat -- reference behaviour)
    double* det_xywh;   // [cap][4] survivors
    double* det_conf;
    int* det_cls;
    int* det_xyxy_i;     // [cap][4]  RectInfo.tolist(): int() truncation
    double* det_xyxy_d;  // same values as fp64 (tracker input)
};

// LDS carve for yolo_post_frame: doubles first (8-byte aligned), then ints.
struct YoloLds {
    double *c0, *c1, *c2, *c3, *sc, *area, *red_v;
    int *idx, *order, *red_i, *scan;
    unsigned char* flag;
    static ADAS_HD size_t bytes(int cap, int nthr) {
        return (size_t)cap * (6 * 8 + 2 * 4) + (size_t)nthr * (8 + 4 + 4) + 1024 + 64;
    }
    ADAS_DEV void carve(void* base, int cap, int nthr) {
        double* d = (double*)base;
        c0 = d; d += cap;
        c1 = d; d += cap;
        c2 = d; d += cap;
        c3 = d; d += cap;
        sc = d; d += cap;
        area = d; d += cap;
        red_v = d; d += nthr;
        int* q = (int*)d;
        idx = q; q += cap;
        order = q; q += cap;
        red_i = q; q += nthr;
        scan = q; q += nthr;
        flag = (unsigned char*)q;  // 1024 bytes
    }
};

// ---------------------------------------------------------------------------
// Reference-mode NMS in one wave: candidate j lives in lane j % 64, register slot j / 64 (N <= 64 * NC).  Same
// selection sort, lossy "swap", IoU arithmetic and comparison order as the block-wide loop in yolo_post_frame, without its
// five workgroup barriers per selected box; row broadcasts are lane shuffles.  keep[] receives the survivors in j order.
// ---------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
template <int NC>
ADAS_DEV double nms_pick(const double (&a)[NC], int k) {
    double r = a[0];
#pragma unroll
    for (int q = 1; q < NC; ++q) r = (k == q) ? a[q] : r;
    return r;
}
template <int NC>
ADAS_DEV int nms_picki(const int (&a)[NC], int k) {
    int r = a[0];
#pragma unroll
    for (int q = 1; q < NC; ++q) r = (k == q) ? a[q] : r;
    return r;
}

template <int NC>
ADAS_DEV void yolo_nms_reference_wave(const Ctx& c, double* lc0, double* lc1, double* lc2, double* lc3, const double* lsc,
                                      double* larea, int* lidx, int N, double iou_thr, int* keep, int* n_keep) {
    if (c.tid < 64) {
        const int lane = c.tid;
        double c0[NC], c1[NC], c2[NC], c3[NC], sc[NC], ar[NC];
        int idx[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int j = lane + 64 * k, jj = j < N ? j : 0;
            c0[k] = lc0[jj]; c1[k] = lc1[jj]; c2[k] = lc2[jj]; c3[k] = lc3[jj];
            sc[k] = lsc[jj]; ar[k] = larea[jj]; idx[k] = lidx[jj];
        }
        for (int i = 0; i < N; ++i) {
            const int pos = i + 1;
            double maxscore;
            int maxpos;
            if (i != N - 1) {
                // Scores are float confidences widened to double (or 0.0), so the low 29 mantissa bits are zero and,
                // being non-negative, the bit patterns order like the values: key = bits | ~j packs "highest score, then
                // lowest index" (np.argmax's first maximum) into one 64-bit maximum.
                unsigned long long key = 0;
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int j = lane + 64 * k;
                    if (j >= pos && j < N) {
                        const unsigned long long kk = (unsigned long long)__double_as_longlong(sc[k]) | (unsigned long long)(~j & 0x1fffffff);
                        key = kk > key ? kk : key;
                    }
                }
                key = wave_max_u64(key);
                maxscore = __longlong_as_double((long long)(key & ~0x1fffffffull));
                maxpos = (int)(~key & 0x1fffffffull);
            } else {
                maxscore = wave_read_f64(nms_pick<NC>(sc, (N - 1) >> 6), (N - 1) & 63);
                maxpos = 0;
            }
            if (i != N - 1 && maxscore == 0.0) break;  // scores are >= 0: nothing can change any more
            const int li = i & 63, ki = i >> 6, lm = maxpos & 63, km = maxpos >> 6;
            const double tscore = wave_read_f64(nms_pick<NC>(sc, ki), li);
            // Row broadcasts come from the LDS copy of the boxes/areas/indices (one uniform-address read each; picking a
            // register slot by a run-time index costs NC selects per value), which lane 0 keeps in step with the "swap".
            double i0, i1, i2, i3, ia;
            if (tscore < maxscore) {  // utils.py:218-231: rows "swapped" through a view (boxes + index copied one way only)
                const double tarea = larea[i];
                i0 = lc0[maxpos]; i1 = lc1[maxpos]; i2 = lc2[maxpos]; i3 = lc3[maxpos];
                ia = larea[maxpos];
                const int mi = lidx[maxpos];
#pragma unroll
                for (int k = 0; k < NC; ++k)
                    if (lane == li && k == ki) {
                        c0[k] = i0; c1[k] = i1; c2[k] = i2; c3[k] = i3;
                        idx[k] = mi; sc[k] = maxscore; ar[k] = ia;
                    }
#pragma unroll
                for (int k = 0; k < NC; ++k)
                    if (lane == lm && k == km) {
                        sc[k] = tscore;
                        ar[k] = tarea;
                    }
                if (lane == 0) {
                    lc0[i] = i0; lc1[i] = i1; lc2[i] = i2; lc3[i] = i3;
                    lidx[i] = mi;
                    larea[i] = ia;
                    larea[maxpos] = tarea;
                }
            } else {
                i0 = lc0[i]; i1 = lc1[i]; i2 = lc2[i]; i3 = lc3[i];
                ia = larea[i];
            }
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int j = lane + 64 * k;
                const bool live = j >= pos && j < N && sc[k] != 0.0;  // a suppressed box can only be "suppressed" again
                if (__ballot(live) == 0) continue;
                if (j >= pos && j < N) {
                    double xx1 = fmax(i1, c1[k]);
                    double yy1 = fmax(i0, c0[k]);
                    double xx2 = fmin(i3, c3[k]);
                    double yy2 = fmin(i2, c2[k]);
                    double w = fmax(0.0, xx2 - xx1 + 1);
                    double h = fmax(0.0, yy2 - yy1 + 1);
                    double inter = w * h;
                    double ovr = inter / (ia + ar[k] - inter);
                    if (ovr > iou_thr) sc[k] = 0.0;  // weight 0 (utils.py:247-251)
                }
            }
        }
        int nk = 0;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int j = lane + 64 * k;
            const bool fl = j < N && sc[k] > 0.001;
            const unsigned long long m = __ballot(fl);
            if (fl) keep[nk + __popcll(m & ((1ull << lane) - 1ull))] = idx[k];
            nk += __popcll(m);
        }
        if (lane == 0) *n_keep = nk;
    }
    c.sync();
}
#endif

// phase timers for scratch builds (ADAS_CFLAGS=-DADAS_YP_PROF, read back with adas_debug_yolo_prof)
#if defined(ADAS_YP_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define YP_MARK_INIT unsigned long long yp_t0_ = wall_clock64()
#define YP_MARK(i)                                        \
    do {                                                  \
        if (c.tid == 0) {                                 \
            unsigned long long t_ = wall_clock64();       \
            atomicAdd(&g_yp_prof[i], t_ - yp_t0_);        \
            yp_t0_ = t_;                                  \
        }                                                 \
    } while (0)
#else
#define YP_MARK_INIT
#define YP_MARK(i)
#endif

ADAS_DEV void yolo_post_frame(const Ctx& c, const YoloPostCfg& cfg, const YoloPostFrame& f, void* lds_base) {
    YoloLds L;
    L.carve(lds_base, cfg.cap, c.nthr);
    const int A = cfg.A, cap = cfg.cap;
    YP_MARK_INIT;

    // ---- phase A: ordered stream compaction of anchors with conf > box_score (anchor order preserved)
    int base = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    {   // four anchors per thread per trip (sub-pass u covers a0 + u*nthr + tid): one ballot each, two barriers per trip
        const int lane = c.tid & 63, wv = c.tid >> 6, nw = (c.nthr + 63) >> 6;  // nw <= 16 (scan[] holds 4*nw wave counts)
        for (int a0 = 0; a0 < A; a0 += 4 * c.nthr) {
            bool fl[4];
            int below[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = a0 + u * c.nthr + c.tid;
                fl[u] = a < A && (double)f.best_conf[a] > cfg.box_score;
                const unsigned long long m = __ballot(fl[u]);
                below[u] = __popcll(m & ((1ull << lane) - 1ull));
                if (lane == 0) L.scan[u * nw + wv] = __popcll(m);
            }
            c.sync();
            int run = base;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int mine = run;
                for (int w = 0; w < nw; ++w) {
                    const int cw = L.scan[u * nw + w];
                    if (w < wv) mine += cw;
                    run += cw;
                }
                const int slot = mine + below[u];
                if (fl[u] && slot < cap) f.cand_anchor[slot] = a0 + u * c.nthr + c.tid;
            }
            base = run;
            c.sync();
        }
    }
#else
    for (int a0 = 0; a0 < A; a0 += 1024) {
        int n = (A - a0 < 1024) ? (A - a0) : 1024;
        ADAS_PAR_FOR(c, j, 0, 1024) L.flag[j] = (j < n && (double)f.best_conf[a0 + j] > cfg.box_score) ? 1 : 0;
        c.sync();
        // each thread owns a contiguous slice of the chunk
        int per = (1024 + c.nthr - 1) / c.nthr;
        int lo = c.tid * per, hi = lo + per;
        if (hi > 1024) hi = 1024;
        int cnt = 0;
        for (int j = lo; j < hi; ++j) cnt += L.flag[j];
        L.scan[c.tid] = cnt;
        c.sync();
        if (c.tid == 0) {
            int run = 0;
            for (int t = 0; t < c.nthr; ++t) {
                int v = L.scan[t];
                L.scan[t] = run;
                run += v;
            }
            L.red_i[0] = run;
        }
        c.sync();
        int slot = base + L.scan[c.tid];
        for (int j = lo; j < hi; ++j)
            if (L.flag[j]) {
                if (slot < cap) f.cand_anchor[slot] = a0 + j;
                ++slot;
            }
        base += L.red_i[0];
        c.sync();
    }
#endif
    const int N = base < cap ? base : cap;
    YP_MARK(0);
    if (c.tid == 0) {
        f.counts[0] = base;
        f.counts[1] = N;
        f.counts[3] = (base > cap) ? 1 : 0;
    }

    // ---- phase B: cxcywh -> xyxy (fp64, yoloDetector.py:132) -> inverse letterbox (utils.py:82-86)
    //      -> NMS prep xywh -> xyxy round trip (utils.py:187), +1 areas (utils.py:211)
    ADAS_PAR_FOR(c, j, 0, N) {
        int a = f.cand_anchor[j];
        double x, y, w, h;
        if (cfg.layout == 0) {
            x = (double)f.head[(size_t)0 * A + a];
            y = (double)f.head[(size_t)1 * A + a];
            w = (double)f.head[(size_t)2 * A + a];
            h = (double)f.head[(size_t)3 * A + a];
        } else {
            const float* r = f.head + (size_t)a * (5 + cfg.nc);
            float xf = r[0], yf = r[1], wf = r[2], hf = r[3];
            if (cfg.layout == 2) yolo_lite_decode(cfg.in_h, cfg.in_w, a, xf, yf, wf, hf);
            x = (double)xf;
            y = (double)yf;
            w = (double)wf;
            h = (double)hf;
        }
        double hw = 0.5 * w, hh = 0.5 * h;
        double x1 = x - hw, y1 = y - hh, x2 = x + hw, y2 = y + hh;
        x1 = (x1 - (double)cfg.pad_w) * cfg.ratio_w;
        x2 = (x2 - (double)cfg.pad_w) * cfg.ratio_w;
        y1 = (y1 - (double)cfg.pad_h) * cfg.ratio_h;
        y2 = (y2 - (double)cfg.pad_h) * cfg.ratio_h;
        double bw = x2 - x1, bh = y2 - y1;
        f.cand_xywh[4 * j + 0] = x1;
        f.cand_xywh[4 * j + 1] = y1;
        f.cand_xywh[4 * j + 2] = bw;
        f.cand_xywh[4 * j + 3] = bh;
        double conf = (double)f.best_conf[a];
        f.cand_conf[j] = conf;
        f.cand_cls[j] = f.best_cls[a];
        double r2 = x1 + bw, r3 = y1 + bh;  // _dets[:,2:4] = _dets[:,0:2] + _dets[:,2:4]
        L.c0[j] = x1;
        L.c1[j] = y1;
        L.c2[j] = r2;
        L.c3[j] = r3;
        L.sc[j] = conf;
        L.idx[j] = j;
        if (cfg.nms_mode == 0)
            L.area[j] = (r3 - y1 + 1) * (r2 - x1 + 1);
        else
            L.area[j] = (r2 - x1) * (r3 - y1);
    }
    c.sync();
    YP_MARK(1);

    int K = 0;
    if (N == 1) {  // utils.py:197-198 / :131-132
        if (c.tid == 0) f.keep[0] = 0;
        K = 1;
#if defined(__HIP_DEVICE_COMPILE__)
    } else if (N > 1 && cfg.nms_mode == 0 && N <= 512) {
        // ---- reference mode, register-resident in one wave (same arithmetic as the block-wide loop below)
        if (N <= 64) yolo_nms_reference_wave<1>(c, L.c0, L.c1, L.c2, L.c3, L.sc, L.area, L.idx, N, cfg.iou_thr, f.keep, &L.red_i[0]);
        else if (N <= 128) yolo_nms_reference_wave<2>(c, L.c0, L.c1, L.c2, L.c3, L.sc, L.area, L.idx, N, cfg.iou_thr, f.keep, &L.red_i[0]);
        else if (N <= 256) yolo_nms_reference_wave<4>(c, L.c0, L.c1, L.c2, L.c3, L.sc, L.area, L.idx, N, cfg.iou_thr, f.keep, &L.red_i[0]);
        else yolo_nms_reference_wave<8>(c, L.c0, L.c1, L.c2, L.c3, L.sc, L.area, L.idx, N, cfg.iou_thr, f.keep, &L.red_i[0]);
        K = L.red_i[0];
        c.sync();
#endif
    } else if (N > 1 && cfg.nms_mode == 0) {
        // ---- reference mode: selection sort with the lossy view-"swap" (Appendix A3)
        for (int i = 0; i < N; ++i) {
            const int pos = i + 1;
            double maxscore;
            int maxpos;
            if (i != N - 1) {
                block_argmax_first(c, L.sc, pos, N, L.red_v, L.red_i, maxscore, maxpos);
            } else {
                maxscore = L.sc[N - 1];
                maxpos = 0;
            }
            if (i != N - 1 && maxscore == 0.0) break;  // scores are >= 0: nothing can change any more
            if (c.tid == 0) {
                double tscore = L.sc[i];
                if (tscore < maxscore) {
                    double tarea = L.area[i];
                    L.c0[i] = L.c0[maxpos];
                    L.c1[i] = L.c1[maxpos];
                    L.c2[i] = L.c2[maxpos];
                    L.c3[i] = L.c3[maxpos];
                    L.idx[i] = L.idx[maxpos];
                    L.sc[i] = L.sc[maxpos];
                    L.sc[maxpos] = tscore;
                    L.area[i] = L.area[maxpos];
                    L.area[maxpos] = tarea;
                }
            }
            c.sync();
            const double i0 = L.c0[i], i1 = L.c1[i], i2 = L.c2[i], i3 = L.c3[i], ia = L.area[i];
            ADAS_PAR_FOR(c, j, pos, N) {
                double xx1 = fmax(i1, L.c1[j]);
                double yy1 = fmax(i0, L.c0[j]);
                double xx2 = fmin(i3, L.c3[j]);
                double yy2 = fmin(i2, L.c2[j]);
                double w = fmax(0.0, xx2 - xx1 + 1);
                double h = fmax(0.0, yy2 - yy1 + 1);
                double inter = w * h;
                double ovr = inter / (ia + L.area[j] - inter);
                if (ovr > cfg.iou_thr) L.sc[j] = 0.0;  // weight 0 (utils.py:247-251)
            }
            c.sync();
        }
        c.sync();
        if (c.tid == 0) {
            int k = 0;
            for (int j = 0; j < N; ++j)
                if (L.sc[j] > 0.001) f.keep[k++] = L.idx[j];
            L.red_i[0] = k;
        }
        c.sync();
        K = L.red_i[0];
        c.sync();
    } else if (N > 1) {
        // ---- greedy mode (utils.py:105-159): order = argsort(scores)[::-1], ties -> higher index first
        ADAS_PAR_FOR(c, j, 0, N) {
            double s = L.sc[j];
            int rank = 0;
            for (int k = 0; k < N; ++k) {
                double t = L.sc[k];
                rank += (t > s || (t == s && k > j)) ? 1 : 0;
            }
            L.order[rank] = j;
        }
        c.sync();
        // sc[] is reused as the alive flag of each *rank position*
        ADAS_PAR_FOR(c, p, 0, N) L.sc[p] = 1.0;
        c.sync();
        for (int p = 0; p < N; ++p) {
            if (L.sc[p] == 0.0) continue;  // uniform: LDS value, no writes to position p in flight
            const int i = L.order[p];
            const double i0 = L.c0[i], i1 = L.c1[i], i2 = L.c2[i], i3 = L.c3[i], ia = L.area[i];
            ADAS_PAR_FOR(c, q, p + 1, N) {
                if (L.sc[q] != 0.0) {
                    int j = L.order[q];
                    double xx1 = fmax(i0, L.c0[j]);
                    double yy1 = fmax(i1, L.c1[j]);
                    double xx2 = fmin(i2, L.c2[j]);
                    double yy2 = fmin(i3, L.c3[j]);
                    double w = fmax(0.0, xx2 - xx1);
                    double h = fmax(0.0, yy2 - yy1);
                    double inter = w * h;
                    double ovr = inter / (ia + L.area[j] - inter);
                    if (!(ovr <= cfg.iou_thr)) L.sc[q] = 0.0;
                }
            }
            c.sync();
        }
        if (c.tid == 0) {
            int k = 0;
            for (int p = 0; p < N; ++p)
                if (L.sc[p] != 0.0) f.keep[k++] = L.order[p];
            L.red_i[0] = k;
        }
        c.sync();
        K = L.red_i[0];
        c.sync();
    }
    if (c.tid == 0) f.counts[2] = K;
    c.sync();
    YP_MARK(2);

    // ---- gather survivors: RectInfo(x,y,w,h,conf,label) + tolist() int truncation
    ADAS_PAR_FOR(c, k, 0, K) {
        int j = f.keep[k];
        double x = f.cand_xywh[4 * j + 0], y = f.cand_xywh[4 * j + 1];
        double w = f.cand_xywh[4 * j + 2], h = f.cand_xywh[4 * j + 3];
        f.det_xywh[4 * k + 0] = x;
        f.det_xywh[4 * k + 1] = y;
        f.det_xywh[4 * k + 2] = w;
        f.det_xywh[4 * k + 3] = h;
        f.det_conf[k] = f.cand_conf[j];
        f.det_cls[k] = f.cand_cls[j];
        double v[4] = {x, y, x + w, y + h};
        for (int q = 0; q < 4; ++q) {
            double t = trunc(v[q]);
            f.det_xyxy_d[4 * k + q] = t;
            f.det_xyxy_i[4 * k + q] = (int)t;
        }
    }
    YP_MARK(3);
}

// ===========================================================================
// UFLDv2 lane decode  (ultrafastLaneDetectorV2.py:114-181, _softmax :15-19)
// ===========================================================================
#define ADAS_UFLD_MAXPTS 128  // >= max(num_cls_row, num_cls_col)

struct UfldCfg {
    int grid_row, cls_row, grid_col, cls_col, lanes;  // 200,72,100,81,4 for CULane
    int img_w, img_h;
    int local_width;           // 1
    const double* row_anchor;  // [cls_row] device
    const double* col_anchor;  // [cls_col]
};

struct UfldFrame {
    const float *loc_row, *loc_col, *exist_row, *exist_col;
    int* lane_cnt;  // [4]   order left-side, left-ego, right-ego, right-side == lane index 0,1,2,3
    int* lane_det;  // [4]
    int* lane_pts;  // [4][ADAS_UFLD_MAXPTS][2]
};

struct UfldLds {
    int* amax;           // [ (cls_row + cls_col) * 4 ]
    unsigned char* val;  // same count
    int* cnt;            // [8]: valid counts row lanes 0..3, col lanes 0..3
    static ADAS_HD size_t bytes(int cr, int cc, int lanes = 4) { return (size_t)(cr + cc) * lanes * 5 + 64; }
};

ADAS_DEV double ufld_expect(const float* loc, int stride, int m, int G, int lw) {
    // 3-tap (2*lw+1) softmax expectation around the argmax, fp32 softmax, fp64 expectation
    int lo = m - lw < 0 ? 0 : m - lw;
    int hi = m + lw > G - 1 ? G - 1 : m + lw;
    float v[8], mx = -FLT_MAX;
    int n = hi - lo + 1;
    for (int t = 0; t < n; ++t) {
        v[t] = loc[(size_t)(lo + t) * stride];
        mx = v[t] > mx ? v[t] : mx;
    }
    float s = 0.f;
    for (int t = 0; t < n; ++t) {
        v[t] = (float)exp((double)(v[t] - mx));  // correctly-rounded fp32 exp
        s = (t == 0) ? v[t] : s + v[t];
    }
    double acc = 0.0;
    for (int t = 0; t < n; ++t) {
        double p = (double)(v[t] / s) * (double)(lo + t);
        acc = (t == 0) ? p : acc + p;
    }
    return acc + 0.5;
}

ADAS_DEV void ufld_decode_frame(const Ctx& c, const UfldCfg& cfg, const UfldFrame& f, void* lds_base) {
    const int R = cfg.cls_row, C = cfg.cls_col, NL = cfg.lanes;
    const int nr = R * NL, nc = C * NL;
    UfldLds L;
    L.amax = (int*)lds_base;
    L.cnt = L.amax + (nr + nc);
    L.val = (unsigned char*)(L.cnt + 16);
    ADAS_PAR_FOR(c, q, 0, 8) L.cnt[q] = 0;
    c.sync();
    // argmax over the grid dimension (first max) + existence argmax over 2 (ties -> 0)
    ADAS_PAR_FOR(c, t, 0, nr + nc) {
        const bool row = t < nr;
        const int q = row ? t : t - nr;
        const int stride = row ? nr : nc;
        const float* loc = (row ? f.loc_row : f.loc_col) + q;
        const int G = row ? cfg.grid_row : cfg.grid_col;
        float best = loc[0];
        int m = 0;
        // eight independent loads in flight per trip (the scan itself stays sequential: first maximum wins)
        int g = 1;
        for (; g + 8 <= G; g += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = loc[(size_t)(g + u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (v[u] > best) {
                    best = v[u];
                    m = g + u;
                }
        }
        for (; g < G; ++g) {
            float v = loc[(size_t)g * stride];
            if (v > best) {
                best = v;
                m = g;
            }
        }
        const float* ex = (row ? f.exist_row : f.exist_col) + q;
        L.amax[t] = m;
        L.val[t] = ex[stride] > ex[0] ? 1 : 0;
    }
    c.sync();
    ADAS_PAR_FOR(c, q, 0, 2 * 4) {  // valid-anchor count per (row | column, lane) for the four decoded lanes (NL >= 4 may be larger)
        const bool row = q < 4;
        const int i = row ? q : q - 4, K = row ? R : C, base = row ? 0 : nr;
        int n = 0;
        for (int k = 0; k < K; ++k) n += L.val[base + k * NL + i];
        L.cnt[(row ? 0 : 4) + i] = n;
    }
    c.sync();
    // lanes: row anchors feed lanes {1,2}; column anchors feed lanes {0,3}
    ADAS_PAR_FOR(c, t, 0, nr + nc) {
        const bool row = t < nr;
        const int q = row ? t : t - nr;
        const int k = q / NL, i = q % NL;
        const bool lane_ok = row ? (i == 1 || i == 2) : (i == 0 || i == 3);
        if (!lane_ok || !L.val[t]) continue;
        const int cnt = L.cnt[(row ? 0 : 4) + i];
        const bool enough = row ? ((double)cnt > (double)R / 2) : ((double)cnt > (double)C / 4);
        if (!enough) continue;
        int slot = 0;  // ordered position among valid anchors of this lane
        for (int kk = 0; kk < k; ++kk) slot += L.val[(row ? 0 : nr) + kk * NL + i];
        const int m = L.amax[t];
        int px, py;
        if (row) {
            double o = ufld_expect(f.loc_row + q, nr, m, cfg.grid_row, cfg.local_width);
            o = o / (double)(cfg.grid_row - 1) * (double)cfg.img_w;
            px = (int)o;
            py = (int)(cfg.row_anchor[k] * (double)cfg.img_h);
        } else {
            double o = ufld_expect(f.loc_col + q, nc, m, cfg.grid_col, cfg.local_width);
            o = o / (double)(cfg.grid_col - 1) * (double)cfg.img_h;
            px = (int)(cfg.col_anchor[k] * (double)cfg.img_w);
            py = (int)o;
        }
        f.lane_pts[(i * ADAS_UFLD_MAXPTS + slot) * 2 + 0] = px;
        f.lane_pts[(i * ADAS_UFLD_MAXPTS + slot) * 2 + 1] = py;
    }
    ADAS_PAR_FOR(c, i, 0, 4) {
        const bool row = (i == 1 || i == 2);
        const int cnt = L.cnt[(row ? 0 : 4) + i];
        const bool enough = row ? ((double)cnt > (double)R / 2) : ((double)cnt > (double)C / 4);
        const int n = enough ? cnt : 0;
        f.lane_cnt[i] = n;
        f.lane_det[i] = n > 2 ? 1 : 0;
    }
}

// ===========================================================================
// UFLD (v1) lane decode  (ultrafastLaneDetector.py:96-139, ModelConfig :16-40)
// one (G+1, K, L) fp32 tensor per frame: softmax over the first G grid cells (scipy.special.softmax
// on float32), expectation sum(prob * (g+1)) in fp64, argmax over all G+1 cells (cell G = "no lane").
// ===========================================================================
struct Ufld1Cfg {
    int G, K, L;             // griding_num, cls_num_per_lane, lanes (4)
    int cfg_w, cfg_h;        // ModelConfig.img_w / img_h
    int in_w, in_h;          // network input (800 x 288)
    int src_w, src_h;        // source frame: w_ratio = src_w / cfg_w, h_ratio = src_h / cfg_h (:80)
    const double* row_anchor;  // [K] device
};

ADAS_DEV void ufld1_decode_frame(const Ctx& c, const Ufld1Cfg& cfg, const float* out, int* lane_cnt, int* lane_det,
                                 int* lane_pts, void* lds_base) {
    const int G = cfg.G, K = cfg.K, NL = cfg.L, stride = K * NL;
    double* loc = (double*)lds_base;  // [K][L], index k' = K-1-r (the [:, ::-1, :] flip of :102)
    int* cnt = (int*)(loc + K * NL);  // [L] nonzero count, [L] positive count
    ADAS_PAR_FOR(c, t, 0, K * NL) {
        const int kf = t / NL, l = t % NL;
        const float* col = out + (size_t)(K - 1 - kf) * NL + l;
        float mx = col[0], best = col[0];
        int am = 0;
        for (int g = 1; g < G; ++g) {
            const float v = col[(size_t)g * stride];
            mx = v > mx ? v : mx;
            if (v > best) {
                best = v;
                am = g;
            }
        }
        if (col[(size_t)G * stride] > best) am = G;  // np.argmax over all G+1 cells, first max (:108)
        float ssum = 0.f;
        for (int g = 0; g < G; ++g) {
            const float e = (float)exp((double)(col[(size_t)g * stride] - mx));  // correctly-rounded fp32 exp
            ssum = (g == 0) ? e : ssum + e;
        }
        double acc = 0.0;
        for (int g = 0; g < G; ++g) {
            const float e = (float)exp((double)(col[(size_t)g * stride] - mx));
            const double pv = (double)(e / ssum) * (double)(g + 1);  // prob * idx (:104-107)
            acc = (g == 0) ? pv : acc + pv;
        }
        loc[t] = (am == G) ? 0.0 : acc;  // :109
    }
    c.sync();
    ADAS_PAR_FOR(c, l, 0, NL) {
        int nz = 0, pos = 0;
        for (int k = 0; k < K; ++k) {
            const double v = loc[k * NL + l];
            nz += (v != 0.0) ? 1 : 0;
            pos += (v > 0.0) ? 1 : 0;
        }
        const bool det = nz > 2;  // :123
        cnt[l] = det ? 1 : 0;
        lane_det[l] = det ? 1 : 0;
        lane_cnt[l] = det ? pos : 0;
    }
    c.sync();
    const double csw = (double)(cfg.in_w - 1) / (double)(G - 1);  // np.linspace(0, input_width-1, G) step (:113-114)
    const double w_ratio = (double)cfg.src_w / (double)cfg.cfg_w, h_ratio = (double)cfg.src_h / (double)cfg.cfg_h;
    ADAS_PAR_FOR(c, t, 0, K * NL) {
        const int kf = t / NL, l = t % NL;
        const double v = loc[t];
        if (!cnt[l] || !(v > 0.0)) continue;
        int slot = 0;
        for (int kk = 0; kk < kf; ++kk) slot += (loc[kk * NL + l] > 0.0) ? 1 : 0;
        const double x = v * csw * (double)cfg.cfg_w / (double)cfg.in_w - 1;                            // :129
        const double y = (double)cfg.cfg_h * (cfg.row_anchor[K - 1 - kf] / (double)cfg.in_h) - 1;      // :130
        lane_pts[(l * ADAS_UFLD_MAXPTS + slot) * 2 + 0] = (int)(x * w_ratio);                           // :131
        lane_pts[(l * ADAS_UFLD_MAXPTS + slot) * 2 + 1] = (int)(y * h_ratio);
    }
}

// ===========================================================================
// EfficientDet: replaces EfficientdetDetector.__process_output (efficientdetDetector.py:67-85) + Scaler.convert_boxes_coordinate
// (utils.py:70-87).  The exported graph already holds the decode and the NMS: its three outputs are boxes (n, 4) xyxy in input
// pixels, class ids (n) and confidences (n).  What the reference does on top is the inverse letterbox IN FLOAT32 (a float32 array
// combined with Python scalars stays float32: x = (x - padw) * ratiow, w = x2 - x1), the `conf < box_score` filter (a float32
// scalar against a Python float compares in double) and the label lookup (host).  Order is kept.
// ===========================================================================
struct EffdetCfg {
    int pad_h, pad_w;
    float ratio_h, ratio_w;  // float32(old / new): what NumPy makes of the Python float when it meets the float32 array
    double box_score;
    int cap;
};
struct EffdetFrame {
    const float* boxes;  // [n][4] x1, y1, x2, y2
    const int* ids;      // [n]
    const float* confs;  // [n]
    int n;
    int* count;          // [1] survivors
    float* xywh;         // [cap][4]
    float* conf;         // [cap]
    int* cls;            // [cap]
    int* xyxy_i;         // [cap][4] RectInfo.tolist(): int() of x, y, x + w, y + h evaluated in float32
};
// lds: int[cap + 1] exclusive prefix of the keep flags
ADAS_DEV void effdet_post_frame(const Ctx& c, const EffdetCfg& cfg, const EffdetFrame& f, int* pre) {
    const int n = f.n < cfg.cap ? f.n : cfg.cap;
    ADAS_PAR_FOR(c, i, 0, n) pre[i + 1] = ((double)f.confs[i] < cfg.box_score) ? 0 : 1;   // `if (conf < box_score): continue` -- NaN is kept, as there
    c.sync();
    if (c.tid == 0) {
        pre[0] = 0;
        for (int i = 0; i < n; ++i) pre[i + 1] += pre[i];
        f.count[0] = pre[n];
    }
    c.sync();
    ADAS_PAR_FOR(c, i, 0, n) {
        if (pre[i + 1] == pre[i]) continue;
        const int k = pre[i];
        const float x1 = (f.boxes[i * 4 + 0] - (float)cfg.pad_w) * cfg.ratio_w;
        const float x2 = (f.boxes[i * 4 + 2] - (float)cfg.pad_w) * cfg.ratio_w;
        const float y1 = (f.boxes[i * 4 + 1] - (float)cfg.pad_h) * cfg.ratio_h;
        const float y2 = (f.boxes[i * 4 + 3] - (float)cfg.pad_h) * cfg.ratio_h;
        const float w = x2 - x1, h = y2 - y1;
        f.xywh[k * 4 + 0] = x1; f.xywh[k * 4 + 1] = y1; f.xywh[k * 4 + 2] = w; f.xywh[k * 4 + 3] = h;
        f.conf[k] = f.confs[i];
        f.cls[k] = f.ids[i];
        f.xyxy_i[k * 4 + 0] = (int)x1; f.xyxy_i[k * 4 + 1] = (int)y1;
        f.xyxy_i[k * 4 + 2] = (int)(x1 + w); f.xyxy_i[k * 4 + 3] = (int)(y1 + h);
    }
}

// ===========================================================================
// EfficientDet in-graph tail: what the exported efficientdet-d0 graph does behind its two head tensors before it hands
// (boxes, class ids, confidences) to EfficientdetDetector.__process_output (efficientdetDetector.py:67-70).  The reference ships
// no such graph (it loads an .onnx through onnxruntime): this restates the published post-processing of the architecture
// (arXiv:1911.09070 section 4 / the public PyTorch implementation's Anchors, BBoxTransform, ClipBoxes and batched NMS):
//   anchors   pyramid levels 3..7 (strides 8..128), per cell 3 octave scales 2^(k/3) x 3 aspect ratios (1,1), (1.4,0.7), (0.7,1.4),
//             side = anchor_scale * stride * scale; centre (stride / 2 + i * stride); rows ordered (level, y, x, scale, ratio)
//   score     sigmoid of the largest class logit (first index on ties), kept when > score_thr
//   box       (dy, dx, dh, dw) against the anchor: centre = d * size + centre_a, size = exp(d) * size_a; xyxy; clipped to
//             [0, W - 1] x [0, H - 1]
//   NMS       greedy by descending score (ties: anchor order), a box suppresses later boxes of the SAME class with IoU > iou_thr;
//             at most max_det survivors
// Arithmetic: anchors and results are float32 values, every expression in between is evaluated in double (the convention of this
// file: decisions do not depend on the host's or the device's float32 libm).
// ===========================================================================
struct EffdetTailCfg {
    int in_h, in_w, nc, cap, max_det;
    double score_thr, iou_thr, anchor_scale;
};
struct EffdetTailFrame {
    const float* reg[5];   // level l: [cells_l * 9][4]
    const float* cls[5];   // level l: [cells_l * 9][nc]
    int* count;            // [2]: survivors, candidates (candidates > cap: overflow, nothing else is written)
    float* boxes;          // [max_det][4] x1, y1, x2, y2 in input pixels
    int* ids;              // [max_det]
    float* confs;          // [max_det]
};
struct EffdetTailLds {     // carved from one LDS block of effdet_tail_lds_bytes(cap, nthr)
    float* score;          // [cap] by candidate (anchor order)
    int* anchor;           // [cap]
    int* cid;              // [cap]
    int* order;            // [cap] candidate index by rank
    float* box;            // [cap][4] by rank
    int* supp;             // [cap] by rank
    int* scan;             // [nthr + 1]
};
ADAS_HD size_t effdet_tail_lds_bytes(int cap, int nthr) { return (size_t)cap * (4 + 4 + 4 + 4 + 16 + 4) + ((size_t)nthr + 2) * 4; }
ADAS_DEV EffdetTailLds effdet_tail_carve(void* lds, int cap) {
    EffdetTailLds L;
    unsigned char* q = (unsigned char*)lds;
    L.box = (float*)q; q += (size_t)cap * 16;
    L.score = (float*)q; q += (size_t)cap * 4;
    L.anchor = (int*)q; q += (size_t)cap * 4;
    L.cid = (int*)q; q += (size_t)cap * 4;
    L.order = (int*)q; q += (size_t)cap * 4;
    L.supp = (int*)q; q += (size_t)cap * 4;
    L.scan = (int*)q;
    return L;
}
// anchor a (global row index) -> level, cell, shape; returns the float32 anchor (y1, x1, y2, x2)
ADAS_DEV void effdet_anchor(const EffdetTailCfg& cfg, int a, int& level, int& row, float out[4]) {
    int base = 0;
    level = 0;
    for (int l = 0; l < 5; ++l) {
        const int n = (cfg.in_h >> (3 + l)) * (cfg.in_w >> (3 + l)) * 9;
        if (a < base + n || l == 4) { level = l; break; }
        base += n;
    }
    row = a - base;
    const int stride = 8 << level, wl = cfg.in_w >> (3 + level);
    const int k = row % 9, cell = row / 9;
    const int y = cell / wl, x = cell - y * wl;
    const double scales[3] = {1.0, 1.2599210498948732, 1.5874010519681994};   // 2^(0/3), 2^(1/3), 2^(2/3)
    const double rx[3] = {1.0, 1.4, 0.7}, ry[3] = {1.0, 0.7, 1.4};
    const double side = cfg.anchor_scale * (double)stride * scales[k / 3];
    const double ax2 = side * rx[k % 3] / 2.0, ay2 = side * ry[k % 3] / 2.0;
    const double cy = (double)stride / 2.0 + (double)y * (double)stride, cx = (double)stride / 2.0 + (double)x * (double)stride;
    out[0] = (float)(cy - ay2); out[1] = (float)(cx - ax2); out[2] = (float)(cy + ay2); out[3] = (float)(cx + ax2);
}
ADAS_DEV void effdet_tail_frame(const Ctx& c, const EffdetTailCfg& cfg, const EffdetTailFrame& f, void* lds) {
    EffdetTailLds L = effdet_tail_carve(lds, cfg.cap);
    int total = 0;
    for (int l = 0; l < 5; ++l) total += (cfg.in_h >> (3 + l)) * (cfg.in_w >> (3 + l)) * 9;
    // ---- candidates in anchor order: chunks of nthr anchors, block-wide exclusive scan of the keep flags
    int n_cand = 0;
    for (int a0 = 0; a0 < total; a0 += c.nthr) {
        const int a = a0 + c.tid;
        int keep = 0, best_c = 0;
        float sc = 0.f;
        if (a < total) {
            int base = 0, level = 0;
            for (int l = 0; l < 5; ++l) {
                const int n = (cfg.in_h >> (3 + l)) * (cfg.in_w >> (3 + l)) * 9;
                if (a < base + n || l == 4) { level = l; break; }
                base += n;
            }
            const float* p = f.cls[level] + (size_t)(a - base) * cfg.nc;
            float best = p[0];
            for (int k = 1; k < cfg.nc; ++k) {
                const float v = p[k];
                if (v > best) { best = v; best_c = k; }
            }
            sc = (float)(1.0 / (1.0 + exp(-(double)best)));
            keep = ((double)sc > cfg.score_thr) ? 1 : 0;
        }
        L.scan[c.tid + 1] = keep;
        c.sync();
        if (c.tid == 0) {
            L.scan[0] = 0;
            for (int i = 0; i < c.nthr; ++i) L.scan[i + 1] += L.scan[i];
        }
        c.sync();
        const int pos = n_cand + L.scan[c.tid];
        if (keep && pos < cfg.cap) {
            L.score[pos] = sc; L.anchor[pos] = a; L.cid[pos] = best_c;
        }
        n_cand += L.scan[c.nthr];
        c.sync();
    }
    if (c.tid == 0) f.count[1] = n_cand;
    if (n_cand > cfg.cap) {   // overflow: the caller reports ADAS_ERR_CAPACITY
        if (c.tid == 0) f.count[0] = 0;
        return;
    }
    // ---- rank by (score descending, anchor ascending): candidates are already in anchor order
    ADAS_PAR_FOR(c, i, 0, n_cand) {
        const float si = L.score[i];
        int r = 0;
        for (int j = 0; j < n_cand; ++j) {
            const float sj = L.score[j];
            r += (sj > si || (sj == si && j < i)) ? 1 : 0;
        }
        L.order[r] = i;
    }
    c.sync();
    // ---- decode + clip, by rank
    ADAS_PAR_FOR(c, r, 0, n_cand) {
        const int i = L.order[r];
        int level, row;
        float an[4];
        effdet_anchor(cfg, L.anchor[i], level, row, an);
        const float* d = f.reg[level] + (size_t)row * 4;
        const double ya = ((double)an[0] + (double)an[2]) / 2.0, xa = ((double)an[1] + (double)an[3]) / 2.0;
        const double ha = (double)an[2] - (double)an[0], wa = (double)an[3] - (double)an[1];
        const double w = exp((double)d[3]) * wa, h = exp((double)d[2]) * ha;
        const double yc = (double)d[0] * ha + ya, xc = (double)d[1] * wa + xa;
        double x1 = xc - w / 2.0, y1 = yc - h / 2.0, x2 = xc + w / 2.0, y2 = yc + h / 2.0;
        x1 = x1 < 0.0 ? 0.0 : x1; y1 = y1 < 0.0 ? 0.0 : y1;
        const double xm = (double)(cfg.in_w - 1), ym = (double)(cfg.in_h - 1);
        x2 = x2 > xm ? xm : x2; y2 = y2 > ym ? ym : y2;
        L.box[r * 4 + 0] = (float)x1; L.box[r * 4 + 1] = (float)y1; L.box[r * 4 + 2] = (float)x2; L.box[r * 4 + 3] = (float)y2;
        L.supp[r] = 0;
    }
    c.sync();
    // ---- greedy per-class NMS in rank order
    int kept = 0;
    for (int r = 0; r < n_cand && kept < cfg.max_det; ++r) {
        if (L.supp[r]) continue;   // uniform: every thread reads the same LDS word after the sync below
        const int i = L.order[r];
        if (c.tid == 0) {
            f.boxes[kept * 4 + 0] = L.box[r * 4 + 0]; f.boxes[kept * 4 + 1] = L.box[r * 4 + 1];
            f.boxes[kept * 4 + 2] = L.box[r * 4 + 2]; f.boxes[kept * 4 + 3] = L.box[r * 4 + 3];
            f.ids[kept] = L.cid[i];
            f.confs[kept] = L.score[i];
        }
        ++kept;
        const double ax1 = L.box[r * 4 + 0], ay1 = L.box[r * 4 + 1], ax2 = L.box[r * 4 + 2], ay2 = L.box[r * 4 + 3];
        const double aarea = (ax2 - ax1) * (ay2 - ay1);
        const int ci = L.cid[i];
        ADAS_PAR_FOR(c, q, r + 1, n_cand) {
            if (L.supp[q] || L.cid[L.order[q]] != ci) continue;
            const double bx1 = L.box[q * 4 + 0], by1 = L.box[q * 4 + 1], bx2 = L.box[q * 4 + 2], by2 = L.box[q * 4 + 3];
            const double iw = (ax2 < bx2 ? ax2 : bx2) - (ax1 > bx1 ? ax1 : bx1), ih = (ay2 < by2 ? ay2 : by2) - (ay1 > by1 ? ay1 : by1);
            const double inter = (iw > 0.0 ? iw : 0.0) * (ih > 0.0 ? ih : 0.0);
            const double uni = aarea + (bx2 - bx1) * (by2 - by1) - inter;
            if (inter / uni > cfg.iou_thr) L.supp[q] = 1;
        }
        c.sync();
    }
    if (c.tid == 0) f.count[0] = kept;
}

}  // namespace adas
