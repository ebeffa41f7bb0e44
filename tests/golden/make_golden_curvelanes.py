#!/usr/bin/env python3
"""Golden vectors for the CurveLanes geometry of the UFLDv2 decoder: the REFERENCE's own
UltrafastLaneDetectorV2.__process_output (ultrafastLaneDetectorV2.py:114-181) run under the import stubs of make_golden.py on
10-lane head tensors of the CurveLanes configuration (configs/curvelanes_res18.py: 200/100 grid cells, 72/41 anchors, 10 lanes)
with ModelConfig(UFLDV2_CURVELANES) -- whose column-anchor table has 81 entries for a 41-anchor head (:47): the reference indexes
it with the head's own k, so only its first 41 values are used; that quirk is part of what is pinned.
Run in the build container only:  python tests/golden/make_golden_curvelanes.py   -> tests/golden/ufld_curve_decode.npz"""
import os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG          # stubs + REF path
from synth import curve_cases, digest


def ref_curve(outputs, W, H):
    from TrafficLaneDetector.ufldDetector.ultrafastLaneDetectorV2 import UltrafastLaneDetectorV2, ModelConfig
    from TrafficLaneDetector.ufldDetector.utils import LaneModelType
    d = object.__new__(UltrafastLaneDetectorV2)
    d.cfg = ModelConfig(LaneModelType.UFLDV2_CURVELANES)
    d.img_width, d.img_height = W, H
    pts, status = d._UltrafastLaneDetectorV2__process_output(outputs, d.cfg)
    return [[(int(p[0]), int(p[1])) for p in lane] for lane in pts], [bool(s) for s in status]


def main():
    MG.install_stubs()
    rec = {}
    for tag, outs, W, H in curve_cases():
        lanes, status = ref_curve(outs, W, H)
        for i in range(4):
            rec[f"{tag}_lane{i}"] = np.asarray(lanes[i], np.int64).reshape(-1, 2)
        rec[f"{tag}_status"] = np.asarray(status, np.bool_)
        rec[f"{tag}_sha1"] = np.array("".join(digest(o) for o in outs))
        print(tag, [len(l) for l in lanes], status)
    np.savez_compressed(os.path.join(HERE, "ufld_curve_decode.npz"), **rec)


if __name__ == "__main__":
    main()
