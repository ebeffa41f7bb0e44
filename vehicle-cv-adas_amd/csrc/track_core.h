// track_core.h -- device-resident ByteTrack: one workgroup per video stream.
//
// Follows ObjectTracker/byteTrack/byteTracker.py:62-185 step by step on a
// fixed-capacity track table kept in HBM (no host round trip per frame):
//   matching.py:34-80,108-116  IoU cost (+ score fusion), fp64, no "+1"
//   matching.py:20-31          lap.lapjv(extend_cost=True, cost_limit=t): exact
//                              LAP; solved here by successive shortest augmenting
//                              paths on rows x (cols + one always-free sink of
//                              cost t), which is the extended (T+D)^2 problem
//                              with its identical dummy rows/cols merged.
//   kalman_filter.py:55-86,155-192,126-153,194-226   initiate / multi_predict / project / update
//   strack.py:61-129           multi_predict, activate, re_activate, update, class vote
//   byteTrack/utils.py:9-69    joint / sub / remove_duplicate_stracks
//   strack.py:53,115           trajectories: the last 30 matched detection boxes of a track (what
//                              DrawTrackedOnFrame draws, byteTracker.py:202-215), a ring per track slot
// Not carried: image crops (strack.py:131-143, needs the host frame); they do not influence ids.
#pragma once
#include "post_core.h"

namespace adas {

enum { BT_NEW = 0, BT_TRACKED = 1, BT_LOST = 2, BT_REMOVED = 3 };
#define ADAS_BT_HIST 8
#define ADAS_BT_TRAJ 30   // LimitedList(30), strack.py:53
enum { BT_ERR_DET_OVERFLOW = 1, BT_ERR_TRACK_OVERFLOW = 2, BT_ERR_HIST_OVERFLOW = 4, BT_ERR_NAN_COST = 8 };

struct BtTrack {
    double mean[8];
    double cov[64];
    double score;
    int track_id, state, is_activated, frame_id, start_frame, tracklet_len, class_id;
    int ever_removed;  // id is a member of the reference's removed_stracks list
    int used, tmp;
    int hist_n;
    int hist_cls[ADAS_BT_HIST];
    int hist_cnt[ADAS_BT_HIST];
    int traj_n;   // boxes ever appended to this track's trajectory (the ring keeps the last ADAS_BT_TRAJ)
    int pad_;
};

struct BtOut {  // compact per-track message (base_track.py:61-72 + strack.py:207-215)
    double tlwh[4];
    double score;
    int track_id, state, is_activated, class_id, frame_id, start_frame, tracklet_len;
    int traj_len;  // len(trajectories): min(appended, 30)
};

struct BtHeader {
    int frame_id, id_count, n_tracked, n_lost, err, pad[3];
};

struct BtParams {
    double track_thresh, det_thresh, match_thresh;
    int max_time_lost;
    int MT, MD;  // capacities: tracks per stream, detections per frame
};

// per-stream views into one HBM allocation
struct BtStream {
    BtHeader* hdr;
    int* tracked;  // [MT] slot ids, list order == reference list order
    int* lost;     // [MT]
    BtTrack* slots;  // [MT]
    double* cost;    // [MT*MD] workspace
    BtOut* out;      // [2*MT]: tracked then lost
    double* traj;    // [MT][ADAS_BT_TRAJ][4]: per slot, ring of the matched detections' tlbr (entry k of the list lives at k % 30)
};

struct BtDet {
    const double* tlbr;  // [nd][4] xyxy
    const double* score;
    const int* cls;
    int nd;
};

// ---------------------------------------------------------------- geometry
ADAS_DEV void bt_track_tlbr(const BtTrack& t, double o[4]) {  // strack.py:151-173
    double w = t.mean[2] * t.mean[3];
    double h = t.mean[3];
    double x1 = t.mean[0] - w / 2, y1 = t.mean[1] - h / 2;
    o[0] = x1; o[1] = y1; o[2] = w + x1; o[3] = h + y1;
}
ADAS_DEV void bt_track_tlwh(const BtTrack& t, double o[4]) {
    double w = t.mean[2] * t.mean[3];
    double h = t.mean[3];
    o[0] = t.mean[0] - w / 2; o[1] = t.mean[1] - h / 2; o[2] = w; o[3] = h;
}
ADAS_DEV void bt_det_tlbr(const double* in, double o[4]) {  // tlbr_to_tlwh then tlwh -> tlbr
    double w = in[2] - in[0], h = in[3] - in[1];
    o[0] = in[0]; o[1] = in[1]; o[2] = w + in[0]; o[3] = h + in[1];
}
ADAS_DEV double bt_iou(const double a[4], const double b[4]) {  // matching.py:34-53
    double xx1 = fmax(a[0], b[0]), yy1 = fmax(a[1], b[1]);
    double xx2 = fmin(a[2], b[2]), yy2 = fmin(a[3], b[3]);
    double w = fmax(0.0, xx2 - xx1), h = fmax(0.0, yy2 - yy1);
    double wh = w * h;
    return wh / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - wh);
}

// ---------------------------------------------------------------- Kalman
#define BT_WPOS (1.0 / 20)
#define BT_WVEL (1.0 / 160)

ADAS_DEV void bt_kf_initiate(BtTrack& t, const double tlwh[4]) {  // kalman_filter.py:55-86
    double h = tlwh[3];
    t.mean[0] = tlwh[0] + tlwh[2] / 2;
    t.mean[1] = tlwh[1] + tlwh[3] / 2;
    t.mean[2] = tlwh[2] / tlwh[3];
    t.mean[3] = h;
    for (int i = 4; i < 8; ++i) t.mean[i] = 0.0;
    double sp = 2 * BT_WPOS * h, sv = 10 * BT_WVEL * h;
    double std[8] = {sp, sp, 1e-2, sp, sv, sv, 1e-5, sv};
    for (int i = 0; i < 64; ++i) t.cov[i] = 0.0;
    for (int i = 0; i < 8; ++i) t.cov[i * 9] = std[i] * std[i];
}

ADAS_DEV void bt_kf_predict(BtTrack& t) {  // strack.py:61-72 + kalman_filter.py:155-192
    if (t.state != BT_TRACKED) t.mean[7] = 0.0;
    double h = t.mean[3];
    double sp = BT_WPOS * h, sv = BT_WVEL * h;
    double q[8] = {sp * sp, sp * sp, 1e-2 * 1e-2, sp * sp, sv * sv, sv * sv, 1e-5 * 1e-5, sv * sv};
    for (int i = 0; i < 4; ++i) t.mean[i] = t.mean[i] + t.mean[i + 4];
    double fp[64];  // F P
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) fp[i * 8 + j] = (i < 4) ? t.cov[i * 8 + j] + t.cov[(i + 4) * 8 + j] : t.cov[i * 8 + j];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            double v = (j < 4) ? fp[i * 8 + j] + fp[i * 8 + j + 4] : fp[i * 8 + j];
            t.cov[i * 8 + j] = (i == j) ? v + q[i] : v;
        }
}

ADAS_DEV void bt_kf_update(BtTrack& t, const double z[4]) {  // kalman_filter.py:126-153,194-226
    double h = t.mean[3];
    double sp = BT_WPOS * h;
    double r[4] = {sp * sp, sp * sp, 1e-1 * 1e-1, sp * sp};
    double S[16], Lc[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) S[i * 4 + j] = t.cov[i * 8 + j] + ((i == j) ? r[i] : 0.0);
    for (int i = 0; i < 16; ++i) Lc[i] = 0.0;
    for (int j = 0; j < 4; ++j) {  // Cholesky, lower
        double d = S[j * 4 + j];
        for (int k = 0; k < j; ++k) d -= Lc[j * 4 + k] * Lc[j * 4 + k];
        d = sqrt(d);
        Lc[j * 4 + j] = d;
        for (int i = j + 1; i < 4; ++i) {
            double s = S[i * 4 + j];
            for (int k = 0; k < j; ++k) s -= Lc[i * 4 + k] * Lc[j * 4 + k];
            Lc[i * 4 + j] = s / d;
        }
    }
    double K[32];  // K[c][r], 8x4 : solve S X = (P H^T)^T, K = X^T
    for (int c = 0; c < 8; ++c) {
        double y[4], x[4];
        for (int i = 0; i < 4; ++i) {
            double s = t.cov[c * 8 + i];
            for (int k = 0; k < i; ++k) s -= Lc[i * 4 + k] * y[k];
            y[i] = s / Lc[i * 4 + i];
        }
        for (int i = 3; i >= 0; --i) {
            double s = y[i];
            for (int k = i + 1; k < 4; ++k) s -= Lc[k * 4 + i] * x[k];
            x[i] = s / Lc[i * 4 + i];
        }
        for (int i = 0; i < 4; ++i) K[c * 4 + i] = x[i];
    }
    double innov[4];
    for (int i = 0; i < 4; ++i) innov[i] = z[i] - t.mean[i];
    for (int c = 0; c < 8; ++c) {
        double s = 0.0;
        for (int i = 0; i < 4; ++i) s += innov[i] * K[c * 4 + i];
        t.mean[c] = t.mean[c] + s;
    }
    double SK[32];  // S K^T : 4x8
    for (int i = 0; i < 4; ++i)
        for (int c = 0; c < 8; ++c) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += S[i * 4 + k] * K[c * 4 + k];
            SK[i * 8 + c] = s;
        }
    for (int a = 0; a < 8; ++a)
        for (int b = 0; b < 8; ++b) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += K[a * 4 + k] * SK[k * 8 + b];
            t.cov[a * 8 + b] = t.cov[a * 8 + b] - s;
        }
}

ADAS_DEV void bt_class_vote(BtTrack& t, int cls, int* err) {  // strack.py:122-129
    int k = -1;
    for (int i = 0; i < t.hist_n; ++i)
        if (t.hist_cls[i] == cls) k = i;
    if (k < 0) {
        if (t.hist_n < ADAS_BT_HIST) {
            k = t.hist_n++;
            t.hist_cls[k] = cls;
            t.hist_cnt[k] = 1;  // history.get(c, 1)
        } else {
            *err |= BT_ERR_HIST_OVERFLOW;
            return;
        }
    }
    t.hist_cnt[k] += 1;
    int best = 0;
    for (int i = 1; i < t.hist_n; ++i)
        if (t.hist_cnt[i] > t.hist_cnt[best]) best = i;  // first max in insertion order
    t.class_id = t.hist_cls[best];
}

// strack.py:88-120: update() when the track is Tracked, re_activate(new_id=False) otherwise
// (`traj`: the slot's trajectory ring.  update() appends new_track.tlbr -- the detection's tlbr -> tlwh -> tlbr round trip, strack.py:115
// -- re_activate() does not, strack.py:88-99.)
ADAS_DEV void bt_apply_match(BtTrack& t, double* traj, const double* det_in, double score, int cls, int fid, bool reactivate, int* err) {
    double w = det_in[2] - det_in[0], h = det_in[3] - det_in[1];
    double z[4] = {det_in[0] + w / 2, det_in[1] + h / 2, w / h, h};
    bt_kf_update(t, z);
    if (!reactivate) {
        bt_det_tlbr(det_in, traj + (size_t)(t.traj_n % ADAS_BT_TRAJ) * 4);
        t.traj_n += 1;
    }
    t.tracklet_len = reactivate ? 0 : t.tracklet_len + 1;
    t.state = BT_TRACKED;
    t.is_activated = 1;
    t.frame_id = fid;
    t.score = score;
    bt_class_vote(t, cls, err);
}

// ---------------------------------------------------------------- LAP
struct LapLds {
    double *u, *v, *minv, *red_v;
    int *way, *p, *red_i;
    unsigned char* used;
};

// argmin over j in [0,n) with !used[j]; ties -> lowest j
ADAS_DEV void block_argmin_unused(const Ctx& c, const double* a, const unsigned char* used, int n, double* red_v,
                                  int* red_i, double& out_v, int& out_i) {
    double bv = DBL_MAX;
    int bi = 0x7fffffff;
    for (int j = c.tid; j < n; j += c.nthr) {
        if (used[j]) continue;
        double v = a[j];
        if (v < bv || bi == 0x7fffffff) {
            bv = v;
            bi = j;
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_down(bv, off, 64);
        int oi = __shfl_down(bi, off, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov < bv || (ov == bv && oi < bi))) {
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
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov < bv || (ov == bv && oi < bi))) {
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

// rows [0,T) x cols [0,D); cost row-major with leading dimension D.  x_row[i] = col or -1, y_col[j] = row or -1.
ADAS_DEV void bt_lap(const Ctx& c, const double* cost, int T, int D, double limit, int* x_row, int* y_col,
                     const LapLds& S) {
    ADAS_PAR_FOR(c, j, 0, D + 1) {
        S.v[j] = 0.0;
        S.p[j] = -1;
    }
    ADAS_PAR_FOR(c, i, 0, T) {
        S.u[i] = 0.0;
        x_row[i] = -1;
    }
    c.sync();
    if (D > 0) {
        for (int r = 0; r < T; ++r) {
            ADAS_PAR_FOR(c, j, 0, D + 1) {
                S.minv[j] = DBL_MAX;
                S.used[j] = 0;
                S.way[j] = -1;
            }
            c.sync();
            int i0 = r, j0 = -1;
            for (;;) {
                const double ui = S.u[i0];
                const double* crow = cost + (size_t)i0 * D;
                ADAS_PAR_FOR(c, j, 0, D + 1) {
                    if (!S.used[j]) {
                        double cur = ((j < D) ? crow[j] : limit) - ui - S.v[j];
                        if (cur < S.minv[j]) {
                            S.minv[j] = cur;
                            S.way[j] = j0;
                        }
                    }
                }
                c.sync();
                double delta;
                int j1;
                block_argmin_unused(c, S.minv, S.used, D + 1, S.red_v, S.red_i, delta, j1);
                ADAS_PAR_FOR(c, j, 0, D + 1) {
                    if (S.used[j]) {
                        S.u[S.p[j]] += delta;
                        S.v[j] -= delta;
                    } else {
                        S.minv[j] -= delta;
                    }
                }
                if (c.tid == 0) S.u[r] += delta;
                c.sync();
                j0 = j1;
                if (j1 == D || S.p[j1] < 0) break;
                if (c.tid == 0) S.used[j1] = 1;
                i0 = S.p[j1];
                c.sync();
            }
            if (c.tid == 0) {  // augment along way[]
                int j = j0;
                for (;;) {
                    int jp = S.way[j];
                    int row = (jp < 0) ? r : S.p[jp];
                    if (j == D) {
                        x_row[row] = -1;
                    } else {
                        S.p[j] = row;
                        x_row[row] = j;
                    }
                    if (jp < 0) break;
                    j = jp;
                }
            }
            c.sync();
        }
    }
    ADAS_PAR_FOR(c, j, 0, D) y_col[j] = S.p[j];
    c.sync();
}

// ---------------------------------------------------------------- LAP, single-wave form
// Same successive-shortest-path iteration as bt_lap, for D + 1 <= 64 * NC columns: wave 0 keeps the per-column state
// (v, minv, way, p, used) in registers, lane l owning columns l, l + 64, ...; the per-step argmin is a 6-step butterfly
// and no workgroup barrier is needed inside the search.  Element arithmetic, comparison order and tie rule (lowest
// column) are those of bt_lap, so the assignment is the same one.
#if defined(__HIP_DEVICE_COMPILE__)
template <int NC>
ADAS_DEV int lap_pick(const int (&a)[NC], int k) {
    int r = a[0];
#pragma unroll
    for (int q = 1; q < NC; ++q) r = (k == q) ? a[q] : r;
    return r;
}

template <int NC>
ADAS_DEV double lap_pickd(const double (&a)[NC], int k) {
    double r = a[0];
#pragma unroll
    for (int q = 1; q < NC; ++q) r = (k == q) ? a[q] : r;
    return r;
}

template <int NC>
ADAS_DEV void bt_lap_wave(const Ctx& c, const double* cost, int T, int D, double limit, int* x_row, int* y_col,
                          const LapLds& S) {
    if (c.tid < 64) {
        const int lane = c.tid;
        double v[NC], minv[NC];
        int way[NC], p[NC], used[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            v[k] = 0.0;
            p[k] = -1;
        }
        for (int i = lane; i < T; i += 64) {
            S.u[i] = 0.0;
            x_row[i] = -1;
        }
        __builtin_amdgcn_wave_barrier();
        if (D > 0) {
            for (int r = 0; r < T; ++r) {
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    minv[k] = DBL_MAX;
                    used[k] = 0;
                    way[k] = -1;
                }
                int i0 = r, j0 = -1;
                for (;;) {
                    const double ui = S.u[i0];
                    const double* crow = cost + (size_t)i0 * D;
                    double lv = HUGE_VAL;  // this lane's smallest slack among its unused columns
#pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        const int j = lane + 64 * k;
                        if (j <= D && !used[k]) {
                            double cur = ((j < D) ? crow[j] : limit) - ui - v[k];
                            if (cur < minv[k]) {
                                minv[k] = cur;
                                way[k] = j0;
                            }
                            lv = minv[k] < lv ? minv[k] : lv;
                        }
                    }
                    // argmin with the lowest column on ties: wave minimum, then the first column that attains it
                    const double gmin = wave_min_f64(lv);
                    int j1 = -1;
#pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        const int j = lane + 64 * k;
                        const unsigned long long m = __ballot(j <= D && !used[k] && minv[k] == gmin);
                        if (j1 < 0 && m) j1 = __builtin_ctzll(m) + 64 * k;
                    }
                    const double delta = wave_read_f64(lap_pickd<NC>(minv, j1 >> 6), j1 & 63);  // the selected element's own value
#pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        const int j = lane + 64 * k;
                        if (j <= D) {
                            if (used[k]) {
                                S.u[p[k]] += delta;
                                v[k] -= delta;
                            } else {
                                minv[k] -= delta;
                            }
                        }
                    }
                    if (lane == 0) S.u[r] += delta;
                    __builtin_amdgcn_wave_barrier();
                    j0 = j1;
                    const int pj1 = wave_read_i32(lap_pick<NC>(p, j1 >> 6), j1 & 63);
                    if (j1 == D || pj1 < 0) break;
#pragma unroll
                    for (int k = 0; k < NC; ++k)
                        if (lane + 64 * k == j1) used[k] = 1;
                    i0 = pj1;
                }
                int j = j0;  // augment along way[]
                for (;;) {
                    const int jp = wave_read_i32(lap_pick<NC>(way, j >> 6), j & 63);
                    const int jq = jp < 0 ? 0 : jp;
                    const int pjp = wave_read_i32(lap_pick<NC>(p, jq >> 6), jq & 63);
                    const int row = (jp < 0) ? r : pjp;
                    if (j == D) {
                        if (lane == 0) x_row[row] = -1;
                    } else {
#pragma unroll
                        for (int k = 0; k < NC; ++k)
                            if (lane + 64 * k == j) p[k] = row;
                        if (lane == 0) x_row[row] = j;
                    }
                    if (jp < 0) break;
                    j = jp;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int j = lane + 64 * k;
            if (j < D) y_col[j] = p[k];
        }
    }
    c.sync();
}
#endif

// dispatcher: the register form whenever the columns fit 8 per lane, the block-wide form otherwise (and on the host)
ADAS_DEV void bt_assign(const Ctx& c, const double* cost, int T, int D, double limit, int* x_row, int* y_col,
                        const LapLds& S) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (D + 1 <= 64) return bt_lap_wave<1>(c, cost, T, D, limit, x_row, y_col, S);
    if (D + 1 <= 128) return bt_lap_wave<2>(c, cost, T, D, limit, x_row, y_col, S);
    if (D + 1 <= 256) return bt_lap_wave<4>(c, cost, T, D, limit, x_row, y_col, S);
    if (D + 1 <= 512) return bt_lap_wave<8>(c, cost, T, D, limit, x_row, y_col, S);
#endif
    bt_lap(c, cost, T, D, limit, x_row, y_col, S);
}

// ---------------------------------------------------------------- ordered compaction
// out[base + rank] = val(k) for every k in [0,n) with keep(k), list order preserved; returns base + count in every
// thread.  keep/val may read anything written before the call's first barrier; out is complete on return.
template <class Keep, class Val>
ADAS_DEV int bt_compact(const Ctx& c, int n, int* out, int base, int* wsum, Keep keep, Val val) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = c.tid & 63, wv = c.tid >> 6, nw = (c.nthr + 63) >> 6;
    for (int k0 = 0; k0 < n; k0 += c.nthr) {
        const int k = k0 + c.tid;
        const bool f = (k < n) && keep(k);
        const unsigned long long m = __ballot(f);
        if (lane == 0) wsum[wv] = __popcll(m);
        c.sync();
        int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        int tot = 0;
        for (int w = 0; w < nw; ++w) {
            const int cw = wsum[w];
            if (w < wv) pos += cw;
            tot += cw;
        }
        if (f) out[pos] = val(k);
        base += tot;
        c.sync();
    }
    return base;
#else
    (void)wsum;
    for (int k = 0; k < n; ++k)
        if (keep(k)) out[base++] = val(k);
    return base;
#endif
}

// ---------------------------------------------------------------- LDS carve for the update
#define ADAS_BT_LDS_BUDGET (156 * 1024)
struct BtLds {
    LapLds lap;
    int *hi, *lo, *rem, *pool, *unc, *rtr, *x, *y, *refind, *lostnew, *newtr, *la, *lb, *mark, *pst, *wsum;
    int* n;  // scalar mailbox [16]
    double* cost;   // association cost matrix when it fits (cost_cap doubles), else BtStream.cost in HBM
    size_t cost_cap;
    static ADAS_HD size_t fixed_bytes(int MT, int MD, int nthr) {
        size_t d = (size_t)MT + 2 * (size_t)(MD + 1) + nthr;           // u, v, minv, red_v
        size_t i = 2 * (size_t)(MD + 1) + nthr                         // way, p, red_i
                   + 4 * (size_t)MD                                    // hi lo rem y
                   + 11 * (size_t)MT + 16 + 16;                        // pool unc rtr x refind lostnew newtr la lb mark pst, n, wsum
        return d * 8 + i * 4 + (size_t)(MD + 1) + 64;
    }
    static ADAS_HD size_t cost_doubles(int MT, int MD, int nthr) {
        size_t f = (fixed_bytes(MT, MD, nthr) + 15) & ~(size_t)15;
        size_t room = f < (size_t)ADAS_BT_LDS_BUDGET ? ((size_t)ADAS_BT_LDS_BUDGET - f) / 8 : 0;
        size_t want = (size_t)MT * MD;
        return want < room ? want : room;
    }
    static ADAS_HD size_t bytes(int MT, int MD, int nthr) {
        return ((fixed_bytes(MT, MD, nthr) + 15) & ~(size_t)15) + cost_doubles(MT, MD, nthr) * 8;
    }
    ADAS_DEV void carve(void* base, int MT, int MD, int nthr) {
        double* d = (double*)base;
        lap.u = d; d += MT;
        lap.v = d; d += MD + 1;
        lap.minv = d; d += MD + 1;
        lap.red_v = d; d += nthr;
        int* q = (int*)d;
        lap.way = q; q += MD + 1;
        lap.p = q; q += MD + 1;
        lap.red_i = q; q += nthr;
        hi = q; q += MD;
        lo = q; q += MD;
        rem = q; q += MD;
        y = q; q += MD;
        pool = q; q += MT;
        unc = q; q += MT;
        rtr = q; q += MT;
        x = q; q += MT;
        refind = q; q += MT;
        lostnew = q; q += MT;
        newtr = q; q += MT;
        la = q; q += MT;
        lb = q; q += MT;
        mark = q; q += MT;
        pst = q; q += MT;
        n = q; q += 16;
        wsum = q; q += 16;
        lap.used = (unsigned char*)q;
        cost = (double*)((unsigned char*)base + ((fixed_bytes(MT, MD, nthr) + 15) & ~(size_t)15));
        cost_cap = cost_doubles(MT, MD, nthr);
    }
};

enum { N_ERR = 0 };

// phase timers for tools/experiments/bt_bench.hip (scratch builds only)
#if defined(ADAS_BT_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define BT_MARK_INIT unsigned long long bt_t0_ = wall_clock64()
#define BT_MARK(i)                                      \
    do {                                                \
        if (c.tid == 0) {                               \
            unsigned long long t_ = wall_clock64();     \
            atomicAdd(&g_bt_prof[i], t_ - bt_t0_);      \
            bt_t0_ = t_;                                \
        }                                               \
    } while (0)
#else
#define BT_MARK_INIT
#define BT_MARK(i)
#endif

// ---------------------------------------------------------------- BYTETracker.update
// List membership tests of the reference compare track ids (byteTrack/utils.py:9-40).  A table slot keeps one id for
// as long as it is in use and ids are never reissued, so "same id" == "same slot" and the tests become per-slot marks
// in LDS; every filtered list is then an ordered compaction over the workgroup (bt_compact) instead of a serial scan.
ADAS_DEV void bytetrack_update(const Ctx& c, const BtParams& P, const BtStream& S, const BtDet& det, void* lds_base) {
    BtLds L;
    L.carve(lds_base, P.MT, P.MD, c.nthr);
    BtTrack* slots = S.slots;
    const int MT = P.MT, MD = P.MD;
    BT_MARK_INIT;
    const int fid = S.hdr->frame_id + 1;
    const int n_tr0 = S.hdr->n_tracked, n_lost0 = S.hdr->n_lost, id0 = S.hdr->id_count;
    int nd = det.nd;
    if (c.tid == 0) L.n[N_ERR] = (nd > MD) ? BT_ERR_DET_OVERFLOW : 0;
    if (nd > MD) nd = MD;
    ADAS_PAR_FOR(c, q, 0, MT) L.mark[q] = 0;
    c.sync();
    if (c.tid == 0) S.hdr->frame_id = fid;
    // byteTracker.py:73-83 score bands (a score of exactly track_thresh is in neither)
    const int n_hi = bt_compact(c, nd, L.hi, 0, L.wsum, [&](int d) { return det.score[d] > P.track_thresh; }, [&](int d) { return d; });
    const int n_lo = bt_compact(c, nd, L.lo, 0, L.wsum,
                                [&](int d) { double s = det.score[d]; return s > 0.1 && s < P.track_thresh; }, [&](int d) { return d; });
    // :93-102 unconfirmed / tracked split, pool = joint(tracked, lost)
    int n_pool = bt_compact(c, n_tr0, L.pool, 0, L.wsum, [&](int k) { return slots[S.tracked[k]].is_activated != 0; },
                            [&](int k) { return S.tracked[k]; });
    const int n_unc = bt_compact(c, n_tr0, L.unc, 0, L.wsum, [&](int k) { return slots[S.tracked[k]].is_activated == 0; },
                                 [&](int k) { return S.tracked[k]; });
    ADAS_PAR_FOR(c, k, 0, n_pool) L.mark[L.pool[k]] = 1;
    c.sync();
    n_pool = bt_compact(c, n_lost0, L.pool, n_pool, L.wsum, [&](int k) { return !L.mark[S.lost[k]]; }, [&](int k) { return S.lost[k]; });
    BT_MARK(0);

    // :104 STrack.multi_predict(strack_pool); pst = state before this frame's updates
    ADAS_PAR_FOR(c, k, 0, n_pool) {
        BtTrack& t = slots[L.pool[k]];
        L.pst[k] = t.state;
        bt_kf_predict(t);
    }
    c.sync();
    BT_MARK(1);

    // :105-108 first association: IoU cost fused with detection score, cost_limit = match_thresh
    double* cm = ((size_t)n_pool * n_hi <= L.cost_cap) ? L.cost : S.cost;
    ADAS_PAR_FOR(c, e, 0, n_pool * n_hi) {
        int i = e / n_hi, j = e % n_hi, d = L.hi[j];
        double a[4], b[4];
        bt_track_tlbr(slots[L.pool[i]], a);
        bt_det_tlbr(det.tlbr + 4 * d, b);
        double cost = 1 - bt_iou(a, b);
        double sim = 1 - cost;
        double v = 1 - sim * det.score[d];
        cm[e] = (v == v) ? v : DBL_MAX;
    }
    c.sync();
    BT_MARK(2);
    bt_assign(c, cm, n_pool, n_hi, P.match_thresh, L.x, L.y, L.lap);
    BT_MARK(3);
    const int n_ref = bt_compact(c, n_pool, L.refind, 0, L.wsum, [&](int i) { return L.x[i] >= 0 && L.pst[i] != BT_TRACKED; },
                                 [&](int i) { return L.pool[i]; });
    const int n_rtr = bt_compact(c, n_pool, L.rtr, 0, L.wsum, [&](int i) { return L.x[i] < 0 && L.pst[i] == BT_TRACKED; },
                                 [&](int i) { return L.pool[i]; });  // :128 r_tracked_stracks
    const int n_rem = bt_compact(c, n_hi, L.rem, 0, L.wsum, [&](int j) { return L.y[j] < 0; },
                                 [&](int j) { return L.hi[j]; });  // :148 detections = [detections[i] for i in u_detection]
    BT_MARK(4);
    ADAS_PAR_FOR(c, i, 0, n_pool) {
        if (L.x[i] >= 0) {
            int s = L.pool[i], d = L.hi[L.x[i]];
            bt_apply_match(slots[s], S.traj + (size_t)s * ADAS_BT_TRAJ * 4, det.tlbr + 4 * d, det.score[d], det.cls[d], fid, L.pst[i] != BT_TRACKED, &L.n[N_ERR]);
        }
    }
    c.sync();
    BT_MARK(5);

    // :122-145 second association: still-Tracked leftovers vs low-score detections, plain IoU, limit 0.5
    cm = ((size_t)n_rtr * n_lo <= L.cost_cap) ? L.cost : S.cost;
    ADAS_PAR_FOR(c, e, 0, n_rtr * n_lo) {
        int i = e / n_lo, j = e % n_lo, d = L.lo[j];
        double a[4], b[4];
        bt_track_tlbr(slots[L.rtr[i]], a);
        bt_det_tlbr(det.tlbr + 4 * d, b);
        double v = 1 - bt_iou(a, b);
        cm[e] = (v == v) ? v : DBL_MAX;
    }
    c.sync();
    BT_MARK(6);
    bt_assign(c, cm, n_rtr, n_lo, 0.5, L.x, L.y, L.lap);
    BT_MARK(7);
    ADAS_PAR_FOR(c, i, 0, n_rtr) {
        if (L.x[i] >= 0) {
            int s = L.rtr[i], d = L.lo[L.x[i]];
            bt_apply_match(slots[s], S.traj + (size_t)s * ADAS_BT_TRAJ * 4, det.tlbr + 4 * d, det.score[d], det.cls[d], fid, false, &L.n[N_ERR]);
        }
    }
    c.sync();
    BT_MARK(8);
    const int n_lostnew = bt_compact(c, n_rtr, L.lostnew, 0, L.wsum,
                                     [&](int i) { return L.x[i] < 0 && slots[L.rtr[i]].state != BT_LOST; }, [&](int i) { return L.rtr[i]; });
    ADAS_PAR_FOR(c, k, 0, n_lostnew) slots[L.lostnew[k]].state = BT_LOST;
    c.sync();
    BT_MARK(9);

    // :147-159 unconfirmed tracks vs remaining high-score detections, fused cost, limit 0.7
    cm = ((size_t)n_unc * n_rem <= L.cost_cap) ? L.cost : S.cost;
    ADAS_PAR_FOR(c, e, 0, n_unc * n_rem) {
        int i = e / n_rem, j = e % n_rem, d = L.rem[j];
        double a[4], b[4];
        bt_track_tlbr(slots[L.unc[i]], a);
        bt_det_tlbr(det.tlbr + 4 * d, b);
        double cost = 1 - bt_iou(a, b);
        double sim = 1 - cost;
        double v = 1 - sim * det.score[d];
        cm[e] = (v == v) ? v : DBL_MAX;
    }
    c.sync();
    BT_MARK(10);
    bt_assign(c, cm, n_unc, n_rem, 0.7, L.x, L.y, L.lap);
    BT_MARK(11);
    ADAS_PAR_FOR(c, i, 0, n_unc) {
        int s = L.unc[i];
        if (L.x[i] >= 0) {
            int d = L.rem[L.x[i]];
            bt_apply_match(slots[s], S.traj + (size_t)s * ADAS_BT_TRAJ * 4, det.tlbr + 4 * d, det.score[d], det.cls[d], fid, false, &L.n[N_ERR]);
        } else {
            slots[s].state = BT_REMOVED;
            slots[s].tmp = -1;  // joins removed_stracks at the end of this frame
        }
    }
    c.sync();
    BT_MARK(12);

    // :162-168 new tracks from unmatched high-score detections: the k-th of them takes the k-th free slot and id
    // id_count + k + 1 (L.lo is free again and holds the candidate detections)
    const int n_cand = bt_compact(c, n_rem, L.lo, 0, L.wsum, [&](int j) { return L.y[j] < 0 && !(det.score[L.rem[j]] < P.det_thresh); },
                                  [&](int j) { return L.rem[j]; });
    const int n_free = bt_compact(c, MT, L.newtr, 0, L.wsum, [&](int q) { return !slots[q].used; }, [&](int q) { return q; });
    const int n_new = n_cand < n_free ? n_cand : n_free;
    ADAS_PAR_FOR(c, k, 0, n_new) {
        const int d = L.lo[k];
        BtTrack& t = slots[L.newtr[k]];
        const double* b = det.tlbr + 4 * d;
        double tlwh[4] = {b[0], b[1], b[2] - b[0], b[3] - b[1]};
        t.used = 1;
        t.track_id = id0 + k + 1;
        bt_kf_initiate(t, tlwh);
        t.score = det.score[d];
        t.tracklet_len = 0;
        t.state = BT_TRACKED;
        t.is_activated = (fid == 1) ? 1 : 0;
        t.frame_id = fid;
        t.start_frame = fid;
        t.class_id = det.cls[d];
        t.hist_n = 1;
        t.hist_cls[0] = det.cls[d];
        t.hist_cnt[0] = 1;
        t.ever_removed = 0;
        t.tmp = 0;
        t.traj_n = 0;
    }
    if (c.tid == 0) {
        S.hdr->id_count = id0 + n_new;
        if (n_cand > n_free) L.n[N_ERR] |= BT_ERR_TRACK_OVERFLOW;
    }
    // :171-174 age out lost tracks
    ADAS_PAR_FOR(c, k, 0, n_lost0) {
        BtTrack& t = slots[S.lost[k]];
        if (fid - t.frame_id > P.max_time_lost) {
            t.state = BT_REMOVED;
            t.tmp = -1;
        }
    }
    ADAS_PAR_FOR(c, q, 0, MT) L.mark[q] = 0;
    c.sync();
    // :176-182 list algebra: tracked = joint(joint(tracked & Tracked, activated), refind)
    int nt = bt_compact(c, n_tr0, L.la, 0, L.wsum, [&](int k) { return slots[S.tracked[k]].state == BT_TRACKED; },
                        [&](int k) { return S.tracked[k]; });
    ADAS_PAR_FOR(c, k, 0, n_new) L.la[nt + k] = L.newtr[k];
    nt += n_new;
    c.sync();
    ADAS_PAR_FOR(c, k, 0, nt) L.mark[L.la[k]] = 1;
    c.sync();
    nt = bt_compact(c, n_ref, L.la, nt, L.wsum, [&](int k) { return !L.mark[L.refind[k]]; }, [&](int k) { return L.refind[k]; });
    ADAS_PAR_FOR(c, k, 0, n_ref) L.mark[L.refind[k]] = 1;
    c.sync();
    // lost = sub(sub(lost, tracked) + newly lost, removed_stracks as of the previous frame)
    int nl = bt_compact(c, n_lost0, L.lb, 0, L.wsum, [&](int k) { int s = S.lost[k]; return !L.mark[s] && !slots[s].ever_removed; },
                        [&](int k) { return S.lost[k]; });
    nl = bt_compact(c, n_lostnew, L.lb, nl, L.wsum, [&](int k) { return !slots[L.lostnew[k]].ever_removed; },
                    [&](int k) { return L.lostnew[k]; });
    ADAS_PAR_FOR(c, q, 0, MT) {
        if (slots[q].used && slots[q].tmp == -1) {
            slots[q].ever_removed = 1;  // removed_stracks.extend(removed)
            slots[q].tmp = 0;
        }
    }
    BT_MARK(13);
    // :183 remove_duplicate_stracks: pairs with IoU distance < 0.15; x/rtr reused as duplicate flags
    ADAS_PAR_FOR(c, k, 0, nt) L.x[k] = 0;
    ADAS_PAR_FOR(c, k, 0, nl) L.rtr[k] = 0;
    c.sync();
    ADAS_PAR_FOR(c, e, 0, nt * nl) {
        int ia = e / nl, ib = e % nl;
        const BtTrack& ta = slots[L.la[ia]];
        const BtTrack& tb = slots[L.lb[ib]];
        double a[4], b[4];
        bt_track_tlbr(ta, a);
        bt_track_tlbr(tb, b);
        double dist = 1 - bt_iou(a, b);
        if (dist < 0.15) {
            int time_a = ta.frame_id - ta.start_frame, time_b = tb.frame_id - tb.start_frame;
            if (time_a > time_b)
                L.rtr[ib] = 1;
            else
                L.x[ia] = 1;
        }
    }
    c.sync();
    BT_MARK(14);
    const int a = bt_compact(c, nt, S.tracked, 0, L.wsum, [&](int k) { return !L.x[k]; }, [&](int k) { return L.la[k]; });
    const int b = bt_compact(c, nl, S.lost, 0, L.wsum, [&](int k) { return !L.rtr[k]; }, [&](int k) { return L.lb[k]; });
    if (c.tid == 0) {
        S.hdr->n_tracked = a;
        S.hdr->n_lost = b;
        S.hdr->err |= L.n[N_ERR];
    }
    // free slots that are in neither list
    ADAS_PAR_FOR(c, q, 0, MT) L.mark[q] = 0;
    c.sync();
    ADAS_PAR_FOR(c, k, 0, a + b) L.mark[k < a ? S.tracked[k] : S.lost[k - a]] = 1;
    c.sync();
    ADAS_PAR_FOR(c, q, 0, MT) {
        const int m = L.mark[q];
        slots[q].tmp = m;
        if (slots[q].used && !m) slots[q].used = 0;
    }
    BT_MARK(15);
    // compact messages: tracked first, then lost
    ADAS_PAR_FOR(c, k, 0, a + b) {
        const BtTrack& t = slots[k < a ? S.tracked[k] : S.lost[k - a]];
        BtOut& o = S.out[k];
        bt_track_tlwh(t, o.tlwh);
        o.score = t.score;
        o.track_id = t.track_id;
        o.state = t.state;
        o.is_activated = t.is_activated;
        o.class_id = t.class_id;
        o.frame_id = t.frame_id;
        o.start_frame = t.start_frame;
        o.tracklet_len = t.tracklet_len;
        o.traj_len = t.traj_n < ADAS_BT_TRAJ ? t.traj_n : ADAS_BT_TRAJ;
    }
    BT_MARK(16);
}

// The trajectories of the tracks in message order (tracked, then lost), oldest box first: lens[k] = len(trajectories),
// out[k][i] = trajectories[i] (tlbr) for i < lens[k].
ADAS_DEV void bytetrack_gather_trajectories(const Ctx& c, const BtStream& S, int* lens, double* out) {
    const int a = S.hdr->n_tracked, b = S.hdr->n_lost;
    ADAS_PAR_FOR(c, e, 0, (a + b) * ADAS_BT_TRAJ) {
        const int k = e / ADAS_BT_TRAJ, i = e - k * ADAS_BT_TRAJ;
        const int slot = k < a ? S.tracked[k] : S.lost[k - a];
        const int n = S.slots[slot].traj_n;
        const int len = n < ADAS_BT_TRAJ ? n : ADAS_BT_TRAJ;
        if (i == 0) lens[k] = len;
        if (i < len) {
            const double* src = S.traj + ((size_t)slot * ADAS_BT_TRAJ + (size_t)((n - len + i) % ADAS_BT_TRAJ)) * 4;
            double* dst = out + (size_t)e * 4;
            dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
        }
    }
}

ADAS_DEV void bytetrack_reset(const Ctx& c, const BtParams& P, const BtStream& S) {  // byteTracker.py:187-200
    ADAS_PAR_FOR(c, q, 0, P.MT) S.slots[q].used = 0;
    if (c.tid == 0) {
        S.hdr->frame_id = 0;
        S.hdr->id_count = 0;
        S.hdr->n_tracked = 0;
        S.hdr->n_lost = 0;
        S.hdr->err = 0;
    }
}

}  // namespace adas
