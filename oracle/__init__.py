"""CPU oracle for the per-frame ADAS inference path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`vehicle-cv-adas_amd/`) imports this directory.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may
import it, and there only as the checker / the timed CPU baseline.

Every function is a from-scratch NumPy (post-processing, tracker) or
torch-CPU fp32 (networks) restatement of the reference algorithm and cites
the reference file:line it follows (paths relative to the reference repo
jason-li-831202/Vehicle-CV-ADAS @ 2024_10_08).

Parity pinning status
---------------------
* post-processing + ByteTrack: PINNED against golden vectors produced by
  running the reference's own Python under import stubs
  (`tests/golden/make_golden.py`, fixtures in `tests/golden/*.npz|*.json`).
* networks (YOLOv8/v5, UFLDv2): PARITY UNPINNED at the network boundary --
  the reference ships neither weights nor (for YOLO) the architecture, and
  onnxruntime is absent.  What is pinned is the I/O layout
  (yoloDetector.py:110-133, model_culane.py:17-23,56-59) and, for UFLDv2,
  the layer structure (model_culane.py:33-63, backbone.py:14-58).
"""
