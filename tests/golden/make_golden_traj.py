#!/usr/bin/env python3
"""Golden STrack trajectories (strack.py:53,115: the 30-deep list of matched detection boxes DrawTrackedOnFrame draws,
byteTracker.py:202-215) from the REFERENCE tracker, run over the scenes already stored in bytetrack.json.gz:

    python tests/golden/make_golden_traj.py        # needs /root/reference (or ADAS_REFERENCE); writes bytetrack_traj.json.gz

For every scene, at a few checkpoint frames and after the last one: for every track of tracked_stracks + lost_stracks (in list order) its
track_id, the trajectory list (tlbr, fp64) and what filter_trajectories(frame 720x1280, pad (10, 10)) keeps of it (indices)."""
import gzip, json, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG


def run(frames, label_ids, checkpoints):
    from ObjectTracker.byteTrack.byteTracker import BYTETracker
    from ObjectTracker.byteTrack.dtypes import BaseTrack, STrack
    BaseTrack.reset_counter()
    STrack.update_crops = lambda self, frame: None
    names = {"car": (0, 0, 255), "person": (0, 255, 0), "truck": (255, 0, 0)}
    lab = list(names.keys())
    trk = BYTETracker(names=names)
    img = np.zeros((720, 1280, 3), np.uint8)
    out = {}
    for k, fr in enumerate(frames):
        ids = [lab[i] for i in fr["ids"]] if label_ids else fr["ids"]
        trk.update(fr["boxes"], fr["scores"], ids, None)
        if k in checkpoints or k == len(frames) - 1:
            rows = []
            for t in list(trk.tracked_stracks) + list(trk.lost_stracks):
                traj = [[float(v) for v in b] for b in t.trajectories]
                kept = t.filter_trajectories(img, (10, 10))
                keep_idx = [i for i, b in enumerate(t.trajectories) if any(b is q for q in kept)]
                rows.append(dict(track_id=int(t.track_id), full=bool(t.trajectories.full()), trajectory=traj, filtered=keep_idx))
            out[str(k)] = rows
    return out


def main():
    MG.install_stubs()
    with gzip.open(os.path.join(HERE, "bytetrack.json.gz"), "rt") as f:
        bt = json.load(f)
    res = {}
    for tag, v in bt.items():
        n = len(v["frames"])
        cps = {n // 3, (2 * n) // 3, min(n - 1, 35)}
        res[tag] = run(v["frames"], v["label_ids"], cps)
        last = res[tag][str(n - 1)]
        print(tag, "checkpoints", sorted(int(k) for k in res[tag]), "tracks at the end", len(last), "longest trajectory", max([len(r["trajectory"]) for r in last] or [0]))
    with open(os.path.join(HERE, "bytetrack_traj.json.gz"), "wb") as raw:       # mtime 0, no file name: the same bytes on every run
        with gzip.GzipFile(filename="", fileobj=raw, mode="wb", mtime=0) as f:
            f.write(json.dumps(res).encode())


if __name__ == "__main__":
    main()
