// TEST SCAFFOLDING ONLY -- never loaded by the product.
// Compiles csrc/post_core.h + csrc/track_core.h with g++ as single-thread host
// code (Ctx{tid=0,nthr=1}) so the *logic* of the device routines can be checked
// against the golden vectors on the GPU-less build container.  Races and the
// wave-shuffle reductions are only exercised by the real -m gpu tests.
#include <cstdlib>
#include <cstring>
#include <vector>
#include "post_core.h"
#include "track_core.h"
#include "lane_core.h"
using namespace adas;

extern "C" {

// scan: per-anchor first-argmax class + conf (device: yolo_scan_* kernels)
static void scan_host(const float* head, int layout, int A, int nc, float* conf, int* cls) {
    for (int a = 0; a < A; ++a) {
        float best = 0; int bi = 0;
        for (int k = 0; k < nc; ++k) {
            float v = layout == 0 ? head[(size_t)(4 + k) * A + a]
                                  : head[(size_t)a * (5 + nc) + 5 + k] * head[(size_t)a * (5 + nc) + 4];
            if (k == 0 || v > best) { best = v; bi = k; }
        }
        conf[a] = best; cls[a] = bi;
    }
}

int emu_yolo_post(const float* head, int layout, int A, int nc, double box_score, double iou, int nms_mode,
                  int pad_h, int pad_w, double ratio_h, double ratio_w, int cap, int in_h, int in_w,
                  int* counts, int* cand_anchor, double* cand_xywh, double* cand_conf, int* cand_cls, int* keep,
                  double* det_xywh, double* det_conf, int* det_cls, int* det_xyxy_i, double* det_xyxy_d) {
    std::vector<float> conf(A); std::vector<int> cls(A);
    scan_host(head, layout, A, nc, conf.data(), cls.data());
    YoloPostCfg cfg{layout, A, nc, box_score, iou, nms_mode, pad_h, pad_w, ratio_h, ratio_w, cap, in_h, in_w};
    YoloPostFrame f{head, conf.data(), cls.data(), counts, cand_anchor, cand_xywh, cand_conf, cand_cls, keep,
                    det_xywh, det_conf, det_cls, det_xyxy_i, det_xyxy_d};
    std::vector<double> lds(YoloLds::bytes(cap, 1) / 8 + 2);
    Ctx c{0, 1};
    yolo_post_frame(c, cfg, f, lds.data());
    return 0;
}

int emu_ufld(const float* loc_row, const float* loc_col, const float* exist_row, const float* exist_col,
             int grid_row, int cls_row, int grid_col, int cls_col, int img_w, int img_h, int lw,
             const double* row_anchor, const double* col_anchor, int* lane_cnt, int* lane_det, int* lane_pts, int num_lanes) {
    UfldCfg cfg{grid_row, cls_row, grid_col, cls_col, num_lanes, img_w, img_h, lw, row_anchor, col_anchor};
    UfldFrame f{loc_row, loc_col, exist_row, exist_col, lane_cnt, lane_det, lane_pts};
    std::vector<double> lds(UfldLds::bytes(cls_row, cls_col, num_lanes) / 8 + 2);
    Ctx c{0, 1};
    ufld_decode_frame(c, cfg, f, lds.data());
    return 0;
}

int emu_effdet(const float* boxes, const int* ids, const float* confs, int n, int pad_h, int pad_w, double ratio_h, double ratio_w,
               double box_score, int cap, int* count, float* xywh, float* conf, int* cls, int* xyxy_i) {
    EffdetCfg cfg{pad_h, pad_w, (float)ratio_h, (float)ratio_w, box_score, cap};
    EffdetFrame f{boxes, ids, confs, n, count, xywh, conf, cls, xyxy_i};
    std::vector<int> pre((size_t)cap + 2);
    Ctx c{0, 1};
    effdet_post_frame(c, cfg, f, pre.data());
    return 0;
}

// level l tensors of ONE frame, contiguous: reg_all = [A][4], cls_all = [A][nc], rows (level, y, x, anchor)
int emu_effdet_tail(const float* reg_all, const float* cls_all, int in_h, int in_w, int nc, int cap, int max_det, double score_thr, double iou_thr,
                    double anchor_scale, int* count, float* boxes, int* ids, float* confs) {
    EffdetTailCfg cfg{in_h, in_w, nc, cap, max_det, score_thr, iou_thr, anchor_scale};
    EffdetTailFrame f;
    size_t row = 0;
    for (int l = 0; l < 5; ++l) {
        f.reg[l] = reg_all + row * 4;
        f.cls[l] = cls_all + row * nc;
        row += (size_t)(in_h >> (3 + l)) * (size_t)(in_w >> (3 + l)) * 9;
    }
    f.count = count; f.boxes = boxes; f.ids = ids; f.confs = confs;
    std::vector<unsigned char> lds(effdet_tail_lds_bytes(cap, 1) + 64);
    Ctx c{0, 1};
    effdet_tail_frame(c, cfg, f, lds.data());
    return 0;
}

int emu_ufld1(const float* out, int G, int K, int cfg_w, int cfg_h, int in_w, int in_h, int src_w, int src_h,
              const double* row_anchor, int* lane_cnt, int* lane_det, int* lane_pts) {
    Ufld1Cfg cfg{G, K, 4, cfg_w, cfg_h, in_w, in_h, src_w, src_h, row_anchor};
    std::vector<double> lds((size_t)K * 4 + 16);
    Ctx c{0, 1};
    ufld1_decode_frame(c, cfg, out, lane_cnt, lane_det, lane_pts, lds.data());
    return 0;
}

int emu_lane_geometry(const int* lane_cnt, const int* lane_det, const int* lane_pts, int img_h, int bird_w, int bird_h, int adjust,
                       const double* M, int* hdr, double* vals, int* area, int* bird) {
    LaneGeomCfg cfg;
    cfg.img_h = img_h; cfg.bird_w = bird_w; cfg.bird_h = bird_h; cfg.adjust = adjust;
    for (int i = 0; i < 9; ++i) cfg.M[i] = M[i];
    std::vector<double> fx(2 * (size_t)img_h);
    std::vector<int> idx(2 * (size_t)img_h);
    LaneGeomFrame f{lane_cnt, lane_det, lane_pts, hdr, vals, area, bird, fx.data(), idx.data()};
    std::vector<double> lds(lane_lds_bytes(img_h, bird_h) / 8 + 2);
    Ctx c{0, 1};
    lane_geometry_frame(c, cfg, f, lds.data());
    return 0;
}

struct EmuBt { BtParams P; BtStream S; std::vector<char> mem; std::vector<double> lds; };

void* emu_bt_create(double track_thresh, double match_thresh, int track_buffer, double frame_rate, int MT, int MD) {
    EmuBt* e = new EmuBt;
    e->P = BtParams{track_thresh, track_thresh + 0.1, match_thresh, (int)(frame_rate / 30.0 * track_buffer), MT, MD};
    size_t sz = sizeof(BtHeader) + 2 * MT * sizeof(int) + MT * sizeof(BtTrack) + (size_t)MT * MD * 8 + 2 * MT * sizeof(BtOut) + 64 +
                (size_t)MT * ADAS_BT_TRAJ * 32;
    e->mem.assign(sz, 0);
    char* p = e->mem.data();
    e->S.hdr = (BtHeader*)p; p += sizeof(BtHeader);
    e->S.slots = (BtTrack*)p; p += MT * sizeof(BtTrack);
    e->S.cost = (double*)p; p += (size_t)MT * MD * 8;
    e->S.out = (BtOut*)p; p += 2 * MT * sizeof(BtOut);
    e->S.tracked = (int*)p; p += MT * sizeof(int);
    e->S.lost = (int*)p; p += MT * sizeof(int);
    p += (8 - ((size_t)p & 7)) & 7;
    e->S.traj = (double*)p;
    e->lds.assign(BtLds::bytes(MT, MD, 1) / 8 + 2, 0.0);
    return e;
}
void emu_bt_destroy(void* h) { delete (EmuBt*)h; }
void emu_bt_reset(void* h) { EmuBt* e = (EmuBt*)h; Ctx c{0, 1}; bytetrack_reset(c, e->P, e->S); }
int emu_bt_update(void* h, const double* tlbr, const double* score, const int* cls, int nd) {
    EmuBt* e = (EmuBt*)h;
    BtDet d{tlbr, score, cls, nd};
    Ctx c{0, 1};
    bytetrack_update(c, e->P, e->S, d, e->lds.data());
    return e->S.hdr->err;
}
// out: header ints [frame_id, id_count, n_tracked, n_lost, err]; recs: BtOut array
void emu_bt_fetch(void* h, int* hdr, void* recs) {
    EmuBt* e = (EmuBt*)h;
    hdr[0] = e->S.hdr->frame_id; hdr[1] = e->S.hdr->id_count; hdr[2] = e->S.hdr->n_tracked;
    hdr[3] = e->S.hdr->n_lost; hdr[4] = e->S.hdr->err;
    memcpy(recs, e->S.out, (size_t)(hdr[2] + hdr[3]) * sizeof(BtOut));
}
// lens[n_tracked + n_lost], out[..][30][4]: STrack.trajectories in message order
void emu_bt_trajectories(void* h, int* lens, double* out) {
    EmuBt* e = (EmuBt*)h;
    Ctx c{0, 1};
    bytetrack_gather_trajectories(c, e->S, lens, out);
}
int emu_sizeof_btout() { return (int)sizeof(BtOut); }
}
