"""Oracle: ByteTrack (two-stage IoU association + Kalman CV model + lifecycle).

TEST INFRASTRUCTURE (see oracle/__init__.py).  NumPy/SciPy restatement of
  ObjectTracker/byteTrack/byteTracker.py:30-51,62-200   (BYTETracker)
  ObjectTracker/byteTrack/matching.py:20-31,34-80,108-116 (linear_assignment, ious, iou_distance, fuse_score)
  ObjectTracker/byteTrack/utils.py:9-69                 (joint/sub/remove_duplicate_stracks)
  ObjectTracker/byteTrack/dtypes/kalman_filter.py:40-53,55-86,126-153,155-192,194-226
  ObjectTracker/byteTrack/dtypes/strack.py:33-215       (STrack)
  ObjectTracker/byteTrack/dtypes/base_track.py:5-72     (BaseTrack, TrackState)

Third-party arithmetic not under /root/reference: `lap.lapjv(cost,
extend_cost=True, cost_limit=t)` (lap, unpinned, requirements.txt:4; call site
matching.py:24).  Restated from lap's published algorithm: the (T+D)^2
extended matrix is filled with cost_limit/2, its lower-right DxT block is 0,
its upper-left TxD block is the cost; an exact LAP is solved on it and
assignments landing in the extension are reported as -1.  The exact solver
here is scipy.optimize.linear_sum_assignment (the optimum is unique barring
ties).  The id counter is per-tracker here (the reference's is process-global,
base_track.py:12,33-36 -- equal for one tracker in a fresh process).
"""
import numpy as np
import scipy.linalg
from scipy.optimize import linear_sum_assignment

NEW, TRACKED, LOST, REMOVED = 0, 1, 2, 3


# ----------------------------------------------------------------- Kalman
class KalmanFilter:
    def __init__(self):
        ndim, dt = 4, 1.0
        self._motion_mat = np.eye(2 * ndim, 2 * ndim)
        for i in range(ndim):
            self._motion_mat[i, ndim + i] = dt
        self._update_mat = np.eye(ndim, 2 * ndim)
        self._std_weight_position = 1.0 / 20
        self._std_weight_velocity = 1.0 / 160

    def initiate(self, measurement):                       # kalman_filter.py:55-86
        mean = np.r_[measurement, np.zeros_like(measurement)]
        h = measurement[3]
        std = [2 * self._std_weight_position * h, 2 * self._std_weight_position * h, 1e-2,
               2 * self._std_weight_position * h, 10 * self._std_weight_velocity * h,
               10 * self._std_weight_velocity * h, 1e-5, 10 * self._std_weight_velocity * h]
        return mean, np.diag(np.square(std))

    def project(self, mean, covariance):                   # :126-153
        std = [self._std_weight_position * mean[3], self._std_weight_position * mean[3], 1e-1,
               self._std_weight_position * mean[3]]
        innovation_cov = np.diag(np.square(std))
        mean = np.dot(self._update_mat, mean)
        covariance = np.linalg.multi_dot((self._update_mat, covariance, self._update_mat.T))
        return mean, covariance + innovation_cov

    def multi_predict(self, mean, covariance):             # :155-192
        std_pos = [self._std_weight_position * mean[:, 3], self._std_weight_position * mean[:, 3],
                   1e-2 * np.ones_like(mean[:, 3]), self._std_weight_position * mean[:, 3]]
        std_vel = [self._std_weight_velocity * mean[:, 3], self._std_weight_velocity * mean[:, 3],
                   1e-5 * np.ones_like(mean[:, 3]), self._std_weight_velocity * mean[:, 3]]
        sqr = np.square(np.r_[std_pos, std_vel]).T
        motion_cov = np.asarray([np.diag(sqr[i]) for i in range(len(mean))])
        mean = np.dot(mean, self._motion_mat.T)
        left = np.dot(self._motion_mat, covariance).transpose((1, 0, 2))
        covariance = np.dot(left, self._motion_mat.T) + motion_cov
        return mean, covariance

    def update(self, mean, covariance, measurement):       # :194-226
        projected_mean, projected_cov = self.project(mean, covariance)
        chol, lower = scipy.linalg.cho_factor(projected_cov, lower=True, check_finite=False)
        gain = scipy.linalg.cho_solve((chol, lower), np.dot(covariance, self._update_mat.T).T,
                                      check_finite=False).T
        innovation = measurement - projected_mean
        new_mean = mean + np.dot(innovation, gain.T)
        new_cov = covariance - np.linalg.multi_dot((gain, projected_cov, gain.T))
        return new_mean, new_cov


# ----------------------------------------------------------------- STrack
class STrack:
    shared_kalman = KalmanFilter()

    def __init__(self, tlwh, score, class_id):             # strack.py:37-56
        self._tlwh = np.asarray(tlwh, dtype=np.float64)
        self.kalman_filter = None
        self.mean = None
        self.covariance = None
        self.is_activated = False
        self.score = score
        self.tracklet_len = 0
        self.class_id = class_id
        self.class_id_history = {class_id: 1}
        self.trajectories = []
        self.track_id = 0
        self.state = NEW
        self.frame_id = 0
        self.start_frame = 0

    @property
    def end_frame(self):
        return self.frame_id

    @staticmethod
    def multi_predict(stracks):                            # :61-72
        if len(stracks) > 0:
            mm = np.asarray([st.mean.copy() for st in stracks])
            mc = np.asarray([st.covariance for st in stracks])
            for i, st in enumerate(stracks):
                if st.state != TRACKED:
                    mm[i][7] = 0
            mm, mc = STrack.shared_kalman.multi_predict(mm, mc)
            for i, (m, c) in enumerate(zip(mm, mc)):
                stracks[i].mean = m
                stracks[i].covariance = c

    def activate(self, kalman_filter, frame_id, next_id):  # :74-86
        self.kalman_filter = kalman_filter
        self.track_id = next_id()
        self.mean, self.covariance = self.kalman_filter.initiate(self.tlwh_to_xyah(self._tlwh))
        self.tracklet_len = 0
        self.state = TRACKED
        if frame_id == 1:
            self.is_activated = True
        self.frame_id = frame_id
        self.start_frame = frame_id

    def re_activate(self, new_track, frame_id):            # :88-99 (new_id=False)
        self.mean, self.covariance = self.kalman_filter.update(
            self.mean, self.covariance, self.tlwh_to_xyah(new_track.tlwh))
        self.tracklet_len = 0
        self.state = TRACKED
        self.is_activated = True
        self.frame_id = frame_id
        self.score = new_track.score
        self.update_class_id(new_track.class_id)

    def update(self, new_track, frame_id):                 # :101-120
        self.frame_id = frame_id
        self.tracklet_len += 1
        self.mean, self.covariance = self.kalman_filter.update(
            self.mean, self.covariance, self.tlwh_to_xyah(new_track.tlwh))
        self.trajectories.append(new_track.tlbr)
        if len(self.trajectories) > 30:
            self.trajectories.pop(0)
        self.state = TRACKED
        self.is_activated = True
        self.score = new_track.score
        self.update_class_id(new_track.class_id)

    def filter_trajectories(self, frame_hw, pad=(0, 0)):   # :145-149 (frame.shape[:2] instead of the frame)
        padh, padw = pad
        return [b for b in self.trajectories if b[0] >= 0 + padw and b[1] >= 0 + padh and b[2] <= frame_hw[1] - padw and b[3] <= frame_hw[0] - padh]

    def update_class_id(self, class_id):                   # :122-129
        self.class_id_history[class_id] = self.class_id_history.get(class_id, 1) + 1
        self.class_id = max(self.class_id_history, key=self.class_id_history.get)

    @property
    def tlwh(self):                                        # :151-162
        if self.mean is None:
            return self._tlwh.copy()
        ret = self.mean[:4].copy()
        ret[2] *= ret[3]
        ret[:2] -= ret[2:] / 2
        return ret

    @property
    def tlbr(self):
        ret = self.tlwh.copy()
        ret[2:] += ret[:2]
        return ret

    @staticmethod
    def tlwh_to_xyah(tlwh):
        ret = np.asarray(tlwh).copy()
        ret[:2] += ret[2:] / 2
        ret[2] /= ret[3]
        return ret

    @staticmethod
    def tlbr_to_tlwh(tlbr):
        ret = np.asarray(tlbr).copy()
        ret[2:] -= ret[:2]
        return ret


# ----------------------------------------------------------------- matching
def ious(b1, b2):                                          # matching.py:34-53
    b2 = np.expand_dims(b2, 0)
    b1 = np.expand_dims(b1, 1)
    xx1 = np.maximum(b1[..., 0], b2[..., 0]); yy1 = np.maximum(b1[..., 1], b2[..., 1])
    xx2 = np.minimum(b1[..., 2], b2[..., 2]); yy2 = np.minimum(b1[..., 3], b2[..., 3])
    w = np.maximum(0.0, xx2 - xx1); h = np.maximum(0.0, yy2 - yy1)
    wh = w * h
    return wh / ((b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1]) +
                 (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1]) - wh)


def iou_distance(atracks, btracks):                        # :55-80
    atlbrs = [t.tlbr for t in atracks]
    btlbrs = [t.tlbr for t in btracks]
    _ious = np.zeros((len(atlbrs), len(btlbrs)), dtype=np.float64)
    if _ious.size > 0:
        _ious = ious(np.ascontiguousarray(atlbrs, dtype=np.float64),
                     np.ascontiguousarray(btlbrs, dtype=np.float64))
    return 1 - _ious


def fuse_score(cost_matrix, detections):                   # :108-116
    if cost_matrix.size == 0:
        return cost_matrix
    iou_sim = 1 - cost_matrix
    det_scores = np.array([d.score for d in detections])
    det_scores = np.expand_dims(det_scores, axis=0).repeat(cost_matrix.shape[0], axis=0)
    return 1 - iou_sim * det_scores


def lapjv_extended(cost, cost_limit):
    """lap.lapjv(cost, extend_cost=True, cost_limit=cost_limit) -> (x, y)."""
    T, D = cost.shape
    n = T + D
    e = np.full((n, n), cost_limit / 2.0, dtype=np.float64)
    e[T:, D:] = 0
    e[:T, :D] = cost
    r, c = linear_sum_assignment(e)
    x = np.full(n, -1, dtype=np.int64); y = np.full(n, -1, dtype=np.int64)
    x[r] = c; y[c] = r
    x = x[:T].copy(); y = y[:D].copy()
    x[x >= D] = -1
    y[y >= T] = -1
    return x, y


def linear_assignment(cost_matrix, thresh):                # :20-31
    if cost_matrix.size == 0:
        return (np.empty((0, 2), dtype=int), tuple(range(cost_matrix.shape[0])),
                tuple(range(cost_matrix.shape[1])))
    x, y = lapjv_extended(cost_matrix, thresh)
    matches = [[ix, mx] for ix, mx in enumerate(x) if mx >= 0]
    return np.asarray(matches), np.where(x < 0)[0], np.where(y < 0)[0]


# ----------------------------------------------------------------- list algebra
def joint_stracks(a, b):                                   # utils.py:9-30
    seen, res = set(), []
    for t in a + b:
        if t.track_id not in seen:
            seen.add(t.track_id)
            res.append(t)
    return res


def sub_stracks(a, b):                                     # :33-51
    tracks = {t.track_id: t for t in a}
    for tid in {t.track_id for t in b}:
        tracks.pop(tid, None)
    return list(tracks.values())


def remove_duplicate_stracks(a, b):                        # :54-69
    pd = iou_distance(a, b)
    pairs = np.where(pd < 0.15)
    da, db = set(), set()
    for ia, ib in zip(*pairs):
        ta = a[ia].frame_id - a[ia].start_frame
        tb = b[ib].frame_id - b[ib].start_frame
        if ta > tb:
            db.add(ib)
        else:
            da.add(ia)
    return ([t for i, t in enumerate(a) if i not in da],
            [t for i, t in enumerate(b) if i not in db])


# ----------------------------------------------------------------- tracker
class BYTETracker:
    def __init__(self, track_thresh=0.5, track_buffer=30, match_thresh=0.8, frame_rate=30):
        self.tracked_stracks, self.lost_stracks, self.removed_stracks = [], [], []
        self.track_thresh = track_thresh
        self.match_thresh = match_thresh
        self.frame_id = 0
        self.det_thresh = track_thresh + 0.1
        self.buffer_size = int(frame_rate / 30.0 * track_buffer)
        self.max_time_lost = self.buffer_size
        self.kalman_filter = KalmanFilter()
        self._count = 0

    def _next_id(self):
        self._count += 1
        return self._count

    def reset(self):                                       # byteTracker.py:187-200
        self.frame_id = 0
        self.tracked_stracks, self.lost_stracks, self.removed_stracks = [], [], []
        self._count = 0

    def update(self, bboxes, scores, class_ids):           # :62-185
        self.frame_id += 1
        activated, refind, lost, removed = [], [], [], []
        bboxes = np.array(bboxes, dtype=np.float64).reshape(-1, 4)
        scores = np.array(scores, dtype=np.float64).reshape(-1)
        class_ids = np.array(class_ids).reshape(-1)
        remain = scores > self.track_thresh
        second = np.logical_and(scores > 0.1, scores < self.track_thresh)
        dets, dets2 = bboxes[remain], bboxes[second]
        sk, s2 = scores[remain], scores[second]
        ck, c2 = class_ids[remain], class_ids[second]
        detections = [STrack(STrack.tlbr_to_tlwh(b), s, c) for b, s, c in zip(dets, sk, ck)]
        unconfirmed, tracked = [], []
        for t in self.tracked_stracks:
            (tracked if t.is_activated else unconfirmed).append(t)
        pool = joint_stracks(tracked, self.lost_stracks)
        STrack.multi_predict(pool)
        dists = fuse_score(iou_distance(pool, detections), detections)
        matches, u_track, u_det = linear_assignment(dists, self.match_thresh)
        for it, idet in matches:
            t, d = pool[it], detections[idet]
            if t.state == TRACKED:
                t.update(d, self.frame_id); activated.append(t)
            else:
                t.re_activate(d, self.frame_id); refind.append(t)
        detections2 = [STrack(STrack.tlbr_to_tlwh(b), s, c) for b, s, c in zip(dets2, s2, c2)]
        r_tracked = [pool[i] for i in u_track if pool[i].state == TRACKED]
        dists = iou_distance(r_tracked, detections2)
        matches, u_track, _ = linear_assignment(dists, 0.5)
        for it, idet in matches:
            t, d = r_tracked[it], detections2[idet]
            if t.state == TRACKED:
                t.update(d, self.frame_id); activated.append(t)
            else:
                t.re_activate(d, self.frame_id); refind.append(t)
        for it in u_track:
            t = r_tracked[it]
            if not t.state == LOST:
                t.state = LOST; lost.append(t)
        detections = [detections[i] for i in u_det]
        dists = fuse_score(iou_distance(unconfirmed, detections), detections)
        matches, u_unc, u_det = linear_assignment(dists, 0.7)
        for it, idet in matches:
            unconfirmed[it].update(detections[idet], self.frame_id)
            activated.append(unconfirmed[it])
        for it in u_unc:
            unconfirmed[it].state = REMOVED; removed.append(unconfirmed[it])
        for inew in u_det:
            t = detections[inew]
            if t.score < self.det_thresh:
                continue
            t.activate(self.kalman_filter, self.frame_id, self._next_id)
            activated.append(t)
        for t in self.lost_stracks:
            if self.frame_id - t.end_frame > self.max_time_lost:
                t.state = REMOVED; removed.append(t)
        self.tracked_stracks = [t for t in self.tracked_stracks if t.state == TRACKED]
        self.tracked_stracks = joint_stracks(self.tracked_stracks, activated)
        self.tracked_stracks = joint_stracks(self.tracked_stracks, refind)
        self.lost_stracks = sub_stracks(self.lost_stracks, self.tracked_stracks)
        self.lost_stracks.extend(lost)
        self.lost_stracks = sub_stracks(self.lost_stracks, self.removed_stracks)
        self.removed_stracks.extend(removed)
        self.tracked_stracks, self.lost_stracks = remove_duplicate_stracks(self.tracked_stracks, self.lost_stracks)
        return self.snapshot()

    def snapshot(self):
        """Per-frame trace used by the golden fixtures and parity tests."""
        def rec(t):
            return dict(track_id=int(t.track_id), state=int(t.state), is_activated=bool(t.is_activated),
                        score=float(t.score), class_id=t.class_id if isinstance(t.class_id, str) else int(t.class_id),
                        start_frame=int(t.start_frame), frame_id=int(t.frame_id),
                        tracklet_len=int(t.tracklet_len), tlwh=[float(v) for v in t.tlwh])
        return dict(frame_id=int(self.frame_id), count=int(self._count),
                    tracked=[rec(t) for t in self.tracked_stracks],
                    lost=[rec(t) for t in self.lost_stracks])
