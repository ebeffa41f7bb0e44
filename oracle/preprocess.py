"""TEST INFRASTRUCTURE (oracle) -- NumPy restatement of the reference's frame pre-processing.

  yolo_prepare_input  <- YoloDetector.__prepare_input (ObjectDetector/yoloDetector.py:96-102) over
                         Scaler.process_image (ObjectDetector/utils.py:42-63) and cv2.dnn.blobFromImage
  ufld_prepare_input  <- UltrafastLaneDetectorV2.__prepare_input (ufldDetector/ultrafastLaneDetectorV2.py:96-112)

PARITY UNPINNED for the resize: cv2 (opencv-python==4.5.4.60, requirements.txt:1) is not installed here and the
reference holds no fixtures.  `cv_resize_linear_u8` restates OpenCV's 8-bit INTER_LINEAR reference arithmetic
(modules/imgproc/src/resize.cpp: float source coordinate (d+0.5)*scale-0.5, coefficients rounded to 11-bit fixed
point, HResizeLinear int32 rows, VResizeLinear ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2).  Everything around
the resize (letterbox geometry, canvas 114, channel swap, scaling and the float32/float64 promotion of the UFLD
normalisation) follows the reference lines cited and is exact for identity-size frames.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import numpy as np


def _coefs(dsize, ssize, horizontal):
    scale = 1.0 / (float(dsize) / float(ssize))                       # resize.cpp: scale = 1 / inv_scale
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if horizontal:
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s >= ssize - 1
        f[hi] = 0; s[hi] = ssize - 1
    c0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    c1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    s0 = np.clip(s, 0, ssize - 1)
    s1 = np.clip(s + 1, 0, ssize - 1)
    return s0, s1, c0, c1


def cv_resize_linear_u8(img, dsize_wh):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for HxWxC uint8."""
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    sh, sw = img.shape[:2]
    if (dh, dw) == (sh, sw):
        return img.copy()
    x0, x1, a0, a1 = _coefs(dw, sw, True)
    y0, y1, b0, b1 = _coefs(dh, sh, False)
    src = img.astype(np.int64)
    h = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]          # (sh, dw, C), scale 2^11
    r0, r1 = h[y0], h[y1]
    v = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def letterbox_image(srcimg, target_hw, keep_ratio=True):
    """Scaler.process_image (utils.py:42-63) -> (canvas u8, new_shape, pad_shape)."""
    th, tw = int(target_hw[0]), int(target_hw[1])
    padh, padw, newh, neww = 0, 0, th, tw
    if keep_ratio and srcimg.shape[0] != srcimg.shape[1]:
        hw_scale = srcimg.shape[0] / srcimg.shape[1]
        if hw_scale > 1:
            newh, neww = th, int(tw / hw_scale)
            padw = int((tw - neww) * 0.5)
        else:
            newh, neww = int(th * hw_scale) + 1, tw
            padh = int((th - newh) * 0.5)
        img = cv_resize_linear_u8(srcimg, (neww, newh))
        canvas = np.full((th, tw, 3), 114, dtype=np.uint8)
        canvas[padh:padh + newh, padw:padw + neww, :] = img
    else:
        canvas = cv_resize_linear_u8(srcimg, (tw, th))
    return canvas, (newh, neww), (padh, padw)


def yolo_prepare_input(srcimg_bgr, target_hw):
    """-> (1,3,H,W) float32: blobFromImage(image, 1/255.0, swapRB=True) multiplies float32 pixels by the double 1/255."""
    canvas, _, _ = letterbox_image(srcimg_bgr, target_hw)
    rgb = canvas[:, :, ::-1].astype(np.float64)
    blob = (rgb * (1.0 / 255.0)).astype(np.float32)
    return np.ascontiguousarray(blob.transpose(2, 0, 1)[None])


def ufld_prepare_input(image_bgr, input_hw, crop_ratio):
    """-> (1,3,H,W) float32 (ultrafastLaneDetectorV2.py:96-112)."""
    ih, iw = int(input_hw[0]), int(input_hw[1])
    img = image_bgr[:, :, ::-1]
    new_size = (iw, int(ih / crop_ratio))
    img_input = cv_resize_linear_u8(np.ascontiguousarray(img), new_size).astype(np.float32)
    img_input = img_input[-ih:, :, :]
    mean = [0.485, 0.456, 0.406]
    std = [0.229, 0.224, 0.225]
    img_input = ((img_input / 255.0 - mean) / std)          # float32 / float -> float32 ; - list, / list -> float64
    img_input = img_input.transpose(2, 0, 1)[np.newaxis]
    return np.ascontiguousarray(img_input.astype(np.float32))
