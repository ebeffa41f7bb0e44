#!/usr/bin/env python3
"""Golden vectors for the EfficientDet wrapper: the REFERENCE's own EfficientdetDetector.__process_output
(ObjectDetector/efficientdetDetector.py:67-85) with its Scaler (utils.py:30-87), run under the import stubs of make_golden.py on
seeded (boxes, ids, confs) triples.  Run in the build container only:
    python tests/golden/make_golden_effdet.py   -> tests/golden/effdet_post.npz"""
import os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG
from synth import effdet_cases, digest


def main():
    MG.install_stubs()
    from ObjectDetector.efficientdetDetector import EfficientdetDetector
    from ObjectDetector.utils import Scaler
    rec = {}
    for tag, boxes, ids, confs, src, inp, thr in effdet_cases():
        det = object.__new__(EfficientdetDetector)
        det.box_score = thr
        det.class_names = ["c%d" % i for i in range(80)]       # ids 80..89 -> IndexError -> "unknown" (:79-82)
        sc = Scaler(inp, True)
        # Scaler.process_image geometry without the pixels (utils.py:42-63)
        padh, padw, newh, neww = 0, 0, inp[0], inp[1]
        if src[0] != src[1]:
            hw = src[0] / src[1]
            if hw > 1:
                newh, neww = inp[0], int(inp[1] / hw)
                padw = int((inp[1] - neww) * 0.5)
            else:
                newh, neww = int(inp[0] * hw) + 1, inp[1]
                padh = int((inp[0] - newh) * 0.5)
        sc._old_shape, sc._new_shape, sc._pad_shape = src, (newh, neww), (padh, padw)
        res = det._EfficientdetDetector__process_output([boxes.copy(), ids.copy(), confs.copy()], sc)
        rec[tag + "_xywh"] = np.asarray([[r.x, r.y, r.width, r.height] for r in res], np.float32).reshape(-1, 4)
        rec[tag + "_conf"] = np.asarray([r.conf for r in res], np.float32)
        rec[tag + "_label"] = np.asarray([r.label for r in res], dtype="U16")
        rec[tag + "_xyxy_int"] = np.asarray([r.tolist() for r in res], np.int64).reshape(-1, 4)
        rec[tag + "_sha1"] = np.array(digest(boxes, ids, confs))
        print(tag, len(boxes), "->", len(res))
    np.savez_compressed(os.path.join(HERE, "effdet_post.npz"), **rec)


if __name__ == "__main__":
    main()
