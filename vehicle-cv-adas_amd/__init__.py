"""MI355X-native per-frame ADAS inference path (detector + lane + NMS + tracker).

Host-side mirror of the reference's engine seam (coreEngine.py) and of its detector / lane /
tracker wrappers, over the C ABI of libadas_hip.so (include/adas_hip.h).  There is no CPU
fallback: every compute entry point raises if the HIP library or a gfx950 device is missing.

The directory name contains a hyphen, so import it with
    importlib.import_module("vehicle-cv-adas_amd")
(tests/conftest.py and __graft_entry__.py alias it as `adas_amd`).
"""
__version__ = "0.1.0"
