#!/usr/bin/env python3
"""Golden vectors for the UFLD (v1) lane decoder: the reference's UltrafastLaneDetector.__process_output
(ultrafastLaneDetector.py:96-139) run unmodified under the import stubs of make_golden.py (scipy.special is real).
Build container only (needs /root/reference):  python tests/golden/make_golden_ufld1.py
Writes tests/golden/ufld1_decode.npz."""
import os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from synth import ufld1_cases, digest  # noqa: E402


def ref_ufld1(head, cfg_name, input_wh, src_wh):
    sys.path.insert(1, os.path.join(MG.REF, "TrafficLaneDetector"))
    import TrafficLaneDetector.ufldDetector.ultrafastLaneDetector as U
    UltrafastLaneDetector, ModelConfig, LaneModelType = U.UltrafastLaneDetector, U.ModelConfig, U.LaneModelType   # the enum class that module compares against
    mt = LaneModelType.UFLD_TUSIMPLE if cfg_name == "tusimple" else LaneModelType.UFLD_CULANE
    det = object.__new__(UltrafastLaneDetector)
    det.cfg = ModelConfig(mt)
    assert det.cfg.griding_num == head.shape[1] - 1
    det.input_width, det.input_height = input_wh
    det.h_ratio, det.w_ratio = src_wh[1] / det.cfg.img_h, src_wh[0] / det.cfg.img_w          # :80
    lanes, status = det._UltrafastLaneDetector__process_output([head], det.cfg)
    return [list(map(tuple, l)) for l in lanes], [bool(s) for s in status]


def main():
    MG.install_stubs()
    out = {}
    cases = ufld1_cases()
    for tag, cfg, head, iwh, swh in cases:
        lanes, status = ref_ufld1(head, cfg, iwh, swh)
        out[tag + "_in_sha1"] = np.asarray(digest(head))
        for li, lane in enumerate(lanes):
            out[f"{tag}_lane{li}"] = np.asarray(lane, np.int64).reshape(-1, 2)
        out[tag + "_status"] = np.asarray(status, np.bool_)
        print(tag, status, [len(l) for l in lanes], lanes[0][:2])
    out["tags"] = np.asarray([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "ufld1_decode.npz"), **out)


if __name__ == "__main__":
    main()
