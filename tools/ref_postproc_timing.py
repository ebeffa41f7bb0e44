#!/usr/bin/env python3
"""CPU timing of the REFERENCE's own post-processing and tracker (its Python, unmodified, under the import stubs of
tests/golden/make_golden.py) on the synthetic workload shapes of bench.py.  Needs /root/reference: runs in the build container only
(the GPU box has no reference tree), so its output is a committed text file, not a field of bench.py's JSON line.

    python tools/ref_postproc_timing.py > profiles/r02/reference_postproc_cpu.txt

What is timed per frame (batch 1, as the reference runs): YoloDetector.__process_output + Scaler.convert_boxes_coordinate +
NMS.fast_soft_nms (un-jitted: numba is absent) + get_nms_results on a (84, 8400) head with ~60 anchors over box_score,
BYTETracker.update on the survivors, UltrafastLaneDetectorV2.__process_output on the four CULane tensors."""
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as MG
import synth


def main():
    MG.install_stubs()
    import torch
    lb = synth.LB720
    head = synth.synth_v8_head(1)
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        r = MG.ref_yolo_chain(head, "YOLOV8", lb, 0.4, 0.45)
    t_det = (time.perf_counter() - t0) / n
    frames = synth.track_scene(3, 20, 60)
    t0 = time.perf_counter()
    MG.ref_track_run(frames)
    t_trk = (time.perf_counter() - t0) / len(frames)
    outs = synth.synth_ufld(1, lanes=((1, 20, 1.2), (2, 190, -1.5)), cols=((0, 30, .5), (3, 80, -.6)))
    t0 = time.perf_counter()
    for _ in range(n):
        MG.ref_ufld(outs, 1280, 720)
    t_lane = (time.perf_counter() - t0) / n
    print("reference post-processing on this container's CPU (%d cores visible, single Python thread), per frame:" % os.cpu_count())
    print("  YOLO decode + inverse letterbox + fast_soft_nms + RectInfo (%d candidates -> %d boxes): %.2f ms" % (len(r["conf"]), len(r["keep"]), t_det * 1e3))
    print("  BYTETracker.update (~18 detections per frame, 60 frames):                     %.2f ms" % (t_trk * 1e3))
    print("  UFLDv2 __process_output + lane area (4 lanes):                                %.2f ms" % (t_lane * 1e3))
    print("  sum: %.2f ms per frame = %.0f frames/s for the post-processing legs alone (networks excluded)" % ((t_det + t_trk + t_lane) * 1e3, 1.0 / (t_det + t_trk + t_lane)))
    print("device path, same legs, 64 frames per launch (profiles/r02/bench_default.json stages): detector post 0.10 ms + tracker 0.04 ms + lane decode 0.03 ms per 64 frames")


if __name__ == "__main__":
    main()
