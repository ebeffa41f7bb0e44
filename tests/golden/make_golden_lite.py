#!/usr/bin/env python3
"""Golden vectors for the YOLOv5-lite head: the reference's YoloLiteParameters.lite_postprocess
(yoloDetector.py:18-49) and the rest of its detection chain, run unmodified under the import stubs of
make_golden.py.  Build container only (needs /root/reference):  python tests/golden/make_golden_lite.py
Writes tests/golden/yolo_lite.npz.

Promotion note: lite_postprocess runs on the float32 head exactly as in production (array-with-Python-scalar
arithmetic is float32 under NumPy 1.22 and 2.x alike); its float32 result is then widened to float64 before
__process_output so that yoloDetector.py:132 runs in fp64 as under the pinned numpy==1.22.1 (SURVEY finding 5),
the obj*cls product staying exact because the synthetic obj/cls values are dyadic.
"""
import os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from synth import lite_cases, digest  # noqa: E402


def ref_lite_chain(head, input_hw, lb, box_score, iou):
    from ObjectDetector.yoloDetector import YoloDetector, YoloLiteParameters
    from ObjectDetector.utils import ObjectModelType, Scaler, NMS
    det = object.__new__(YoloDetector)
    YoloLiteParameters.__init__(det, ObjectModelType.YOLOV5_LITE, [1, 3, input_hw[0], input_hw[1]], 80)
    assert det.lite
    det.model_type = ObjectModelType.YOLOV5_LITE
    det.box_score, det.box_nms_iou = box_score, iou
    det.class_names = [str(i) for i in range(80)]
    decoded = det.lite_postprocess(head.copy())              # float32, in place on the copy
    assert decoded.dtype == np.float32
    det.lite = False                                         # already decoded; the rest of the chain in the pinned env's fp64
    boxes, cids, confs, _ = det._YoloDetector__process_output(decoded.astype(np.float64))
    sc = Scaler(lb["target"], True)
    sc._old_shape, sc._new_shape, sc._pad_shape = lb["old"], lb["new"], lb["pad"]
    tb = np.asarray(sc.convert_boxes_coordinate(boxes), np.float64).reshape(-1, 4)
    keep = NMS.fast_soft_nms(tb, confs, iou, dets_type="xywh")
    keep_alt = NMS.fast_nms(tb, confs, iou, "xywh")
    infos = det.get_nms_results(tb, confs, cids, np.array([]))
    return dict(decoded_sha1=np.asarray(digest(decoded[:, :4].copy())), decoded_every50=decoded[::50, :4].copy(),
                raw_boxes=np.asarray(boxes, np.float64).reshape(-1, 4),
                cls=np.asarray(cids, np.int64), conf=np.asarray(confs, np.float64), xywh=tb,
                keep=np.asarray(keep, np.int64), keep_alt=np.asarray(keep_alt, np.int64),
                rect_xywh=np.asarray([[r.x, r.y, r.width, r.height] for r in infos], np.float64).reshape(-1, 4),
                rect_conf=np.asarray([r.conf for r in infos], np.float64),
                rect_label=np.asarray([int(r.label) for r in infos], np.int64),
                rect_xyxy_int=np.asarray([r.tolist() for r in infos], np.int64).reshape(-1, 4))


def main():
    MG.install_stubs()
    out = {}
    cases = lite_cases()
    for tag, head, hw, lb, bs, iou in cases:
        r = ref_lite_chain(head, hw, lb, bs, iou)
        out[tag + "_head_sha1"] = np.asarray(digest(head))
        for k, v in r.items():
            out[f"{tag}_{k}"] = v
        print(tag, "cands", len(r["conf"]), "keep", len(r["keep"]), "alt", len(r["keep_alt"]))
    out["tags"] = np.asarray([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "yolo_lite.npz"), **out)


if __name__ == "__main__":
    main()
