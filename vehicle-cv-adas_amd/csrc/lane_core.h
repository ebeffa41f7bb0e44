// lane_core.h -- ego-lane geometry on the device, one workgroup per frame, straight from the
// lane decoder's device-resident points:
//   ufldDetector/core.py:143-158              __update_lanes_status / __update_lanes_area (area = left + flipped right)
//   ufldDetector/core.py:102-141              __adjust_lanes_points (degree-2 np.polyfit of both ego lanes, resampled
//                                             on np.linspace(miny, maxy, image_height))
//   perspectiveTransformation.py:120-142      transformToBirdViewPoints (3x3 homography, int truncation)
//   perspectiveTransformation.py:145-214      calcCurveAndOffset (direction, curvature radius, lateral offset)
// All arithmetic is IEEE fp64 in the reference's operation order.  np.polyfit solves its column-scaled Vandermonde
// system with LAPACK's SVD least squares; here the same scaled system goes through Householder QR.  Both are backward
// stable, so the coefficients agree to ~1e-12 relative -- the integer truncation of a resampled point can differ only
// when the value sits within that distance of an integer (tests allow 1 px on a handful of points).
// Like post_core.h this header also compiles for the host with one thread (tests/hostemu).
#pragma once
#include "track_core.h"  // Ctx, ADAS_PAR_FOR, bt_compact

namespace adas {

#define ADAS_LANE_MAXPTS 128  // == ADAS_UFLD_MAXPTS
enum { LANE_DIR_NONE = 0, LANE_DIR_L = 1, LANE_DIR_R = 2, LANE_DIR_F = 3 };

struct LaneGeomCfg {
    int img_h;           // source frame height: the resampling count of __adjust_lanes_points
    int bird_w, bird_h;  // bird-view image (PerspectiveTransformation.img_size)
    int adjust;          // LaneDetectBase.adjust_lanes
    double M[9];         // frontal -> bird-view homography, row-major
};

// per-frame views
struct LaneGeomFrame {
    const int* lane_cnt;  // [4]  decoder output
    const int* lane_det;  // [4]
    const int* lane_pts;  // [4][ADAS_LANE_MAXPTS][2]
    int* hdr;             // [8]: area_status, n_left, n_right, direction, bird_cnt[4]
    double* vals;         // [2]: curvature radius (m), offset from the lane centre (m)
    int* area;            // [2*img_h][2]: left points, then the right points in reverse order (np.vstack((l, np.flipud(r))))
    int* bird;            // [4][ADAS_LANE_MAXPTS][2]
    double* fx;           // [2][img_h] resampled x of both ego lanes (work area)
    int* idx;             // [2][img_h] kept sample indices (work area)
};

// rows of the least-squares work area one fit needs, and the LDS bytes of a frame (two fits side by side)
ADAS_HD int lane_fit_rows(int img_h, int bird_h) {
    int n = img_h > bird_h ? img_h : bird_h;
    return n > ADAS_LANE_MAXPTS ? n : ADAS_LANE_MAXPTS;
}
ADAS_HD size_t lane_lds_bytes(int img_h, int bird_h) { return 64 + 16 * 8 + (size_t)2 * 4 * lane_fit_rows(img_h, bird_h) * 8; }

// A fit is worked on by one wave (row i belongs to lane i % 64); on the host build the team is a single thread.
struct LaneTeam {
    int lane, width;
};
ADAS_DEV double lane_team_sum(const LaneTeam& t, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)t;
    const unsigned long long r = wave_allreduce_u64((unsigned long long)__double_as_longlong(v), [](unsigned long long a, unsigned long long b) {
        return (unsigned long long)__double_as_longlong(__longlong_as_double((long long)a) + __longlong_as_double((long long)b));
    });
    return __longlong_as_double((long long)r);  // a + b == b + a: every lane of the butterfly ends with the same bits
#else
    (void)t;
    return v;
#endif
}
ADAS_DEV void lane_team_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// least-squares parabola v ~ c0 t^2 + c1 t + c2 through n >= 3 points (np.polyfit(t, v, 2)); W: 4n doubles (LDS)
template <class GetT, class GetV>
ADAS_DEV void lane_polyfit2(const LaneTeam& tm, int n, GetT t_of, GetV v_of, double* W, double c[3]) {
    double part[3] = {0.0, 0.0, 0.0};
    for (int i = tm.lane; i < n; i += tm.width) {  // scale = sqrt((lhs * lhs).sum(axis=0))
        const double t = t_of(i), t2 = t * t;
        part[0] += t2 * t2;
        part[1] += t * t;
        part[2] += 1.0;
    }
    double sc[3];
    for (int k = 0; k < 3; ++k) sc[k] = sqrt(lane_team_sum(tm, part[k]));
    for (int i = tm.lane; i < n; i += tm.width) {
        const double t = t_of(i);
        W[4 * i + 0] = (t * t) / sc[0];
        W[4 * i + 1] = t / sc[1];
        W[4 * i + 2] = 1.0 / sc[2];
        W[4 * i + 3] = v_of(i);
    }
    lane_team_sync();
    double R[3][4];  // rows of the triangular factor and the transformed right-hand side, identical in every lane
    for (int k = 0; k < 3; ++k) {  // Householder reflections, applied to the trailing columns and the right-hand side
        double tail = 0.0;
        for (int i = tm.lane; i < n; i += tm.width)
            if (i > k) tail += W[4 * i + k] * W[4 * i + k];
        tail = lane_team_sum(tm, tail);
        const double akk = W[4 * k + k];
        const double norm = sqrt(akk * akk + tail);
        const double alpha = akk > 0.0 ? -norm : norm;
        const double v0 = akk - alpha;
        const double vn2 = v0 * v0 + tail;
        R[k][k] = alpha;
        for (int j = k + 1; j < 4; ++j) {
            double dot = 0.0;
            for (int i = tm.lane; i < n; i += tm.width)
                if (i > k) dot += W[4 * i + k] * W[4 * i + j];
            dot = lane_team_sum(tm, dot) + v0 * W[4 * k + j];
            const double f = vn2 > 0.0 ? 2.0 * dot / vn2 : 0.0;
            R[k][j] = W[4 * k + j] - f * v0;
            for (int i = tm.lane; i < n; i += tm.width)
                if (i > k) W[4 * i + j] -= f * W[4 * i + k];
        }
        lane_team_sync();
    }
    double x[3];
    for (int k = 2; k >= 0; --k) {
        double s = R[k][3];
        for (int j = k + 1; j < 3; ++j) s -= R[k][j] * x[j];
        x[k] = s / R[k][k];
    }
    for (int k = 0; k < 3; ++k) c[k] = x[k] / sc[k];
}

ADAS_DEV double lane_poly(const double c[3], double y) { return c[0] * (y * y) + c[1] * y + c[2]; }  // f0*y**2 + f1*y + f2

ADAS_DEV void lane_geometry_frame(const Ctx& c, const LaneGeomCfg& cfg, const LaneGeomFrame& f, void* lds_base) {
    const int H = cfg.img_h;
    const int* PL = f.lane_pts + 1 * ADAS_LANE_MAXPTS * 2;  // left ego  (lanes_points[1])
    const int* PR = f.lane_pts + 2 * ADAS_LANE_MAXPTS * 2;  // right ego (lanes_points[2])
    const int nL = f.lane_cnt[1], nR = f.lane_cnt[2];
    const bool area_ok = f.lane_det[1] && f.lane_det[2];     // core.py:143-148
    int* wsum = (int*)lds_base;                              // bt_compact scratch [16]
    double* fit = (double*)((unsigned char*)lds_base + 64);  // [2][3] coefficients, [6..9] bounds / radii
    double* Wq[2];
    Wq[0] = fit + 16;
    Wq[1] = Wq[0] + (size_t)4 * lane_fit_rows(H, cfg.bird_h);
    double* fx = f.fx;
#if defined(__HIP_DEVICE_COMPILE__)
    const int team = c.tid >> 6;
    const LaneTeam tm{c.tid & 63, 64};
    const bool fitter = c.tid < 128;
#else
    const LaneTeam tm{0, 1};
#endif

    // ---- area polygon (core.py:150-158)
    int n_left = 0, n_right = 0;
    if (area_ok) {
        const bool refit = cfg.adjust && nL > 10 && nR > 10;  // :106-121 (either lane with <= 10 points: both returned as they are)
        if (refit) {
#if defined(__HIP_DEVICE_COMPILE__)
            if (fitter) {
                const int* P = team ? PR : PL;
                lane_polyfit2(tm, team ? nR : nL, [&](int i) { return (double)P[2 * i + 1]; }, [&](int i) { return (double)P[2 * i]; }, Wq[team],
                              fit + 3 * team);
            }
#else
            for (int team = 0; team < 2; ++team) {
                const int* P = team ? PR : PL;
                lane_polyfit2(tm, team ? nR : nL, [&](int i) { return (double)P[2 * i + 1]; }, [&](int i) { return (double)P[2 * i]; }, Wq[team],
                              fit + 3 * team);
            }
#endif
            if (c.tid < 2) {
                const int* P = c.tid ? PR : PL;
                const int n = c.tid ? nR : nL;
                int mn = P[1], mx = P[1];
                for (int i = 1; i < n; ++i) {
                    mn = P[2 * i + 1] < mn ? P[2 * i + 1] : mn;
                    mx = P[2 * i + 1] > mx ? P[2 * i + 1] : mx;
                }
                fit[8 + 2 * c.tid] = (double)mn;
                fit[9 + 2 * c.tid] = (double)mx;
            }
#if !defined(__HIP_DEVICE_COMPILE__)
            {   // the single host thread also plays tid 1
                int mn = PR[1], mx = PR[1];
                for (int i = 1; i < nR; ++i) {
                    mn = PR[2 * i + 1] < mn ? PR[2 * i + 1] : mn;
                    mx = PR[2 * i + 1] > mx ? PR[2 * i + 1] : mx;
                }
                fit[10] = (double)mn;
                fit[11] = (double)mx;
            }
#endif
            c.sync();
            const int minL = (int)fit[8], maxL = (int)fit[9], minR = (int)fit[10], maxR = (int)fit[11];
            int maxy = H - 1, miny = H / 3;                                        // :122-123
            maxy = maxy > maxL ? maxy : maxL; miny = miny < minL ? miny : minL;    // :124-129
            maxy = maxy > maxR ? maxy : maxR; miny = miny < minR ? miny : minR;
            const double step = (double)(maxy - miny) / (double)(H - 1);           // np.linspace(miny, maxy, H)
            auto fity = [&](int i) { return (i == H - 1) ? (double)maxy : (double)i * step + (double)miny; };
            ADAS_PAR_FOR(c, i, 0, H) {
                const double y = fity(i);
                fx[i] = lane_poly(fit, y);
                fx[H + i] = lane_poly(fit + 3, y);
            }
            c.sync();
            // :134-135 keep (int(x), int(y)) where y >= min(lane ys) and x >= 0
            n_left = bt_compact(c, H, f.idx, 0, wsum, [&](int i) { return fity(i) >= (double)minL && fx[i] >= 0.0; }, [&](int i) { return i; });
            n_right = bt_compact(c, H, f.idx + H, 0, wsum, [&](int i) { return fity(i) >= (double)minR && fx[H + i] >= 0.0; },
                                 [&](int i) { return i; });
            ADAS_PAR_FOR(c, k, 0, n_left + n_right) {
                const bool left = k < n_left;
                const int i = left ? f.idx[k] : f.idx[H + (n_right - 1 - (k - n_left))];  // np.flipud(right)
                f.area[2 * k] = (int)fx[left ? i : H + i];
                f.area[2 * k + 1] = (int)fity(i);
            }
        } else {
            n_left = nL;
            n_right = nR;
            ADAS_PAR_FOR(c, k, 0, nL + nR) {
                const int* p = k < nL ? PL + 2 * k : PR + 2 * (nR - 1 - (k - nL));
                f.area[2 * k] = p[0];
                f.area[2 * k + 1] = p[1];
            }
        }
    }
    c.sync();

    // ---- bird-view points of all four lanes (perspectiveTransformation.py:120-142)
    ADAS_PAR_FOR(c, e, 0, 4 * ADAS_LANE_MAXPTS) {
        const int l = e / ADAS_LANE_MAXPTS, k = e % ADAS_LANE_MAXPTS;
        if (k < f.lane_cnt[l]) {
            const double x = (double)f.lane_pts[2 * e], y = (double)f.lane_pts[2 * e + 1];
            const double* M = cfg.M;
            // np.einsum('kl,...l->...k') sums its three products as (p0 + p2) + p1 (two-lane SIMD sum of products,
            // observed with the NumPy the goldens were generated under); the int truncation below makes the order visible
            const double X = (M[0] * x + M[2] * 1.0) + M[1] * y;
            const double Y = (M[3] * x + M[5] * 1.0) + M[4] * y;
            const double Z = (M[6] * x + M[8] * 1.0) + M[7] * y;
            f.bird[2 * e] = (int)(X / Z);
            f.bird[2 * e + 1] = (int)(Y / Z);
        }
    }
    c.sync();

    // ---- curvature and offset from the two ego lanes in the bird view (:145-214)
    int direction = LANE_DIR_NONE;
    const int* BL = f.bird + 1 * ADAS_LANE_MAXPTS * 2;
    const int* BR = f.bird + 2 * ADAS_LANE_MAXPTS * 2;
    const int Hb = cfg.bird_h;
    const bool curve = nL >= 3 && nR >= 3 && Hb > 719;
    if (curve) {
        const double ym = 30.0 / 720, xm = 3.7 / 700;  // :183-184
        auto one = [&](int team) {
            const int* B = team ? BR : BL;
            double* cf = fit + 3 * team;
            lane_polyfit2(tm, team ? nR : nL, [&](int i) { return (double)B[2 * i + 1]; }, [&](int i) { return (double)B[2 * i]; }, Wq[team], cf);
            lane_team_sync();
            double cr[3];  // second fit in world space over ploty = 0 .. Hb-1
            lane_polyfit2(tm, Hb, [&](int i) { return (double)i * ym; }, [&](int i) { return lane_poly(cf, (double)i) * xm; }, Wq[team], cr);
            const double y_eval = (double)(Hb - 1);
            const double s = 2 * cr[0] * y_eval * ym + cr[1];
            if (tm.lane == 0) fit[12 + team] = pow(1 + s * s, 1.5) / fabs(2 * cr[0]);  // :190-191
        };
#if defined(__HIP_DEVICE_COMPILE__)
        if (fitter) one(team);
#else
        one(0);
        one(1);
#endif
    }
    c.sync();
    if (c.tid == 0) {
        if (curve) {
            const double* lf = fit;
            const double* rf = fit + 3;
            const double bend = fabs(lf[0]) > fabs(rf[0]) ? lf[0] : rf[0];  // :165-168
            if (bend < -0.00015 && BL[0] <= BL[2 * (nL / 2)]) direction = LANE_DIR_L;
            else if (bend > 0.00015 && BR[0] >= BR[2 * (nR / 2)]) direction = LANE_DIR_R;
            else direction = LANE_DIR_F;
            const double lx = lane_poly(lf, 719.0), rx = lane_poly(rf, 719.0);  // row 719, as the reference hard-codes
            const double lane_width = fabs(lx - rx);
            const double veh_pos = (lx + rx) / 2.;
            f.vals[0] = (fit[12] + fit[13]) / 2;
            f.vals[1] = (veh_pos - (double)cfg.bird_w / 2.) * (3.7 / lane_width);
        } else {
            f.vals[0] = 0.0;
            f.vals[1] = 0.0;
        }
        f.hdr[0] = area_ok ? 1 : 0;
        f.hdr[1] = n_left;
        f.hdr[2] = n_right;
        f.hdr[3] = direction;
        for (int l = 0; l < 4; ++l) f.hdr[4 + l] = f.lane_cnt[l];
    }
}

}  // namespace adas
