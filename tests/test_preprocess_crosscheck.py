"""CPU: independent cross-checks of the OpenCV INTER_LINEAR restatement (oracle/preprocess.cv_resize_linear_u8).

cv2 is absent from this image (parity with opencv-python==4.5.4.60 stays UNPINNED), but the restatement can at least be held
against two implementations that share OpenCV's sampling convention (source coordinate (d + 0.5) * scale - 0.5, no antialiasing,
edge clamp) and differ from it only in arithmetic:
  * torch.nn.functional.interpolate(mode="bilinear", align_corners=False, antialias=False): float32 bilinear;
  * Pillow's Image.resize(BILINEAR) for pure UPSCALES (its filter support grows with the scale factor on downscales, so only
    upscales are comparable): 8-bit fixed point with its own rounding.
OpenCV's 8-bit path rounds the two 11-bit coefficient products (resize.cpp: >>4, >>16, +2, >>2); against exact bilinear that is at
most 1 LSB.  Any indexing, clamping or coefficient mistake in the restatement would show up as a much larger error here."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import preprocess


def float_bilinear(img, dw, dh):
    t = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
    y = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)
    return y[0].permute(1, 2, 0).numpy()


def frames(seed, h, w):
    rng = np.random.default_rng(seed)
    base = np.repeat(np.repeat(rng.integers(0, 255, ((h + 7) // 8, (w + 7) // 8, 3)), 8, 0), 8, 1)[:h, :w]
    return ((base + rng.integers(0, 255, (h, w, 3))) // 2).astype(np.uint8)


@pytest.mark.parametrize("src,dst", [((720, 1280), (361, 640)),      # Scaler.process_image for a 1280x720 frame (utils.py:42-63)
                                     ((720, 1280), (533, 1600)),     # ultrafastLaneDetectorV2.py:97-101 (CULane: 320 / 0.6)
                                     ((1080, 1920), (361, 640)), ((480, 640), (640, 853)), ((37, 53), (91, 17)), ((64, 64), (64, 64))])
def test_restatement_within_one_lsb_of_float_bilinear(src, dst):
    img = frames(src[0] * 7 + dst[1], *src)
    got = preprocess.cv_resize_linear_u8(img, (dst[1], dst[0])).astype(np.int64)
    want = float_bilinear(img, dst[1], dst[0])
    d = np.abs(got - want)
    assert got.shape == (dst[0], dst[1], 3)
    assert d.max() <= 1.0 + 1e-3, d.max()                  # fixed-point rounding only
    # the two truncating shifts (>>4, >>16) before the final rounding make ~12 % of the pixels the second-nearest integer
    assert (d > 0.5 + 1e-3).mean() < 0.2 and d.mean() < 0.3
    if src == dst:
        np.testing.assert_array_equal(got, img)


@pytest.mark.parametrize("src,dst", [((360, 640), (720, 1280)), ((100, 150), (333, 444)), ((48, 64), (49, 65))])
def test_restatement_vs_pillow_on_upscales(src, dst):
    Image = pytest.importorskip("PIL.Image")
    img = frames(3, *src)
    got = preprocess.cv_resize_linear_u8(img, (dst[1], dst[0])).astype(np.int64)
    pil = np.asarray(Image.fromarray(img).resize((dst[1], dst[0]), Image.BILINEAR)).astype(np.int64)
    assert np.abs(got - pil).max() <= 1


def test_letterbox_geometry_and_normalisation_against_plain_numpy():
    """Everything around the resize is exact arithmetic on known values: canvas 114, pad / new shape of utils.py:42-63, channel
    swap and 1/255 scaling (yoloDetector.py:100); ImageNet normalisation with the float32 -> float64 promotion (:104-110)."""
    img = frames(9, 720, 1280)
    canvas, new, pad = preprocess.letterbox_image(img, (640, 640))
    assert new == (361, 640) and pad == (139, 0)
    assert (canvas[:139] == 114).all() and (canvas[139 + 361:] == 114).all()
    x = preprocess.yolo_prepare_input(img, (640, 640))
    np.testing.assert_array_equal(x[0, 0], (canvas[:, :, 2].astype(np.float64) * (1.0 / 255.0)).astype(np.float32))
    sq = frames(10, 533, 1600)                     # already at the resized geometry: the resize is the identity
    y = preprocess.ufld_prepare_input(sq, (320, 1600), 0.6)
    rgb = sq[-320:, :, ::-1].astype(np.float32)
    want = ((rgb / 255.0 - [0.485, 0.456, 0.406]) / [0.229, 0.224, 0.225]).transpose(2, 0, 1).astype(np.float32)
    np.testing.assert_array_equal(y[0], want)
