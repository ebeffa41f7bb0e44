"""Adapters that give the HIP library the same call shape as tests/emu_api.py."""
import numpy as np
from conftest import load_pkg
import importlib

load_pkg()
L = importlib.import_module("adas_amd._lib")
PP = importlib.import_module("adas_amd.postproc")


def yolo_post(head, layout, lb, box_score, iou, nms_mode=0, cap=1024, batch_copies=1, input_hw=None):
    head = np.ascontiguousarray(head, np.float32)
    if layout == 0:
        nc, A = head.shape[0] - 4, head.shape[1]
    else:
        A, nc = head.shape[0], head.shape[1] - 5
    yp = PP.YoloPost(layout, A, nc, box_score, iou, lb, nms_mode, cap, max_batch=batch_copies, input_hw=input_hw)
    try:
        res = yp.run_host(np.stack([head] * batch_copies))
    finally:
        yp.close()
    return res if batch_copies > 1 else res[0]


def ufld(outs, cfg, W, H, lw=1):
    lr, lc = outs[0], outs[1]
    ud = PP.UfldDecode(lr.shape[1], lr.shape[2], lc.shape[1], lc.shape[2], W, H, cfg.row_anchor, cfg.col_anchor, lw, num_lanes=lr.shape[3])
    try:
        return ud.run_host(outs)[0]
    finally:
        ud.close()


def ufld1(head, cfg, input_wh, src_wh):
    ud = PP.Ufld1Decode(cfg.griding_num, cfg.cls_num_per_lane, cfg.img_w, cfg.img_h, input_wh[0], input_wh[1], src_wh[0], src_wh[1],
                        cfg.row_anchor)
    try:
        return ud.run_host(np.asarray(head, np.float32).reshape(1, cfg.griding_num + 1, cfg.cls_num_per_lane, 4))[0]
    finally:
        ud.close()


def track_snapshot(hdr, tracked, lost):
    def rec(r):
        return dict(track_id=int(r["track_id"]), state=int(r["state"]), is_activated=bool(r["is_activated"]),
                    score=float(r["score"]), class_id=int(r["class_id"]), start_frame=int(r["start_frame"]),
                    frame_id=int(r["frame_id"]), tracklet_len=int(r["tracklet_len"]),
                    tlwh=[float(v) for v in r["tlwh"]])
    return dict(frame_id=int(hdr.frame_id), count=int(hdr.id_count), tracked=[rec(r) for r in tracked],
                lost=[rec(r) for r in lost])
