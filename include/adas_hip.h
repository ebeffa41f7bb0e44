/* adas_hip.h -- C ABI of libadas_hip.so: the MI355X-native per-frame ADAS inference path.
 *
 * The reference (jason-li-831202/Vehicle-CV-ADAS @ 2024_10_08) has no FFI: its engine seam is the
 * Python class protocol of coreEngine.py.  Each entry point below names the reference interface it
 * replaces (file:line, relative to the reference repo).  The ctypes stubs a reference maintainer
 * would add are shown in INTEGRATION.md; the in-tree host mirror is vehicle-cv-adas_amd/.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * adas_status; adas_last_error() gives the message of the calling thread's last failure.
 * `stream` is a hipStream_t passed as void* (NULL = the library's own stream).  Pointers named d_*
 * are device (HBM) pointers, h_* host pointers.  All objects are single-threaded like the
 * reference engines (coreEngine.py:94-116): one caller at a time per handle.
 */
#ifndef ADAS_HIP_H
#define ADAS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ADAS_OK = 0,
    ADAS_ERR_INVALID = -1,   /* bad argument */
    ADAS_ERR_IO = -2,        /* model file missing / unreadable (coreEngine.py:12-13) */
    ADAS_ERR_FORMAT = -3,    /* not a model container this library understands (coreEngine.py:14) */
    ADAS_ERR_HIP = -4,       /* HIP runtime failure */
    ADAS_ERR_CAPACITY = -5,  /* a fixed capacity (candidates, tracks, detections) was exceeded */
    ADAS_ERR_NO_DEVICE = -6  /* no gfx950 device visible: the library never falls back to the CPU */
} adas_status;

const char* adas_last_error(void);
int adas_version(void);
/* Number of visible HIP devices (<=0: none) and selection of the device subsequent handles live on
 * (reference: cuda.Device(0) hard-coded at coreEngine.py:47). */
int adas_device_count(void);
int adas_set_device(int index);
/* PCI address ("0000:c1:00.0") of device `index`, so that a one-process-per-GPU launcher can pin its launching thread to the GPU's
 * NUMA node (sharding.pin_rank; the reference is one process on cuda.Device(0), coreEngine.py:47). */
int adas_device_pci_bus_id(int index, char* out, int out_len);
/* Plain device-memory helpers so a ctypes/cgo caller can stage buffers without another runtime. */
int adas_malloc(void** d_ptr, size_t bytes);
int adas_free(void* d_ptr);
int adas_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int adas_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
int adas_synchronize(void);
/* Page-locked host memory for frames that adas_pipeline_step_frames_host uploads asynchronously. */
int adas_host_alloc(void** h_ptr, size_t bytes);
int adas_host_free(void* h_ptr);
/* A pair of timing events for measuring a stretch of work ON THE STREAM IT IS LAUNCHED ON (bench.py's roofline / post_hbm legs:
 * no reference counterpart; the reference times with time.time() around synchronous calls, demo.py:262-281).
 * start/stop record on `stream`; elapsed_ms waits for the stop event. */
typedef struct adas_timer adas_timer;
int adas_timer_create(adas_timer** out);
int adas_timer_destroy(adas_timer* t);
int adas_timer_start(adas_timer* t, void* stream);
int adas_timer_stop(adas_timer* t, void* stream);
int adas_timer_elapsed_ms(adas_timer* t, float* ms);

/* ===================================================================================
 * Engine: replaces EngineBase / OnnxEngine / TensorRTEngine (coreEngine.py:7-39,120-186)
 * =================================================================================== */
typedef struct adas_engine adas_engine;

#define ADAS_PREC_BF16 0 /* bf16 storage + bf16 MFMA, fp32 accumulate (bench precision) */
#define ADAS_PREC_FP32 1 /* fp32 storage + fp32 MFMA (parity precision, 1e-3 vs the fp32 oracle) */
#define ADAS_PREC_FP16 2 /* IEEE half storage + f16 MFMA (bf16's rate, 11 significant bits), fp32 accumulate: the precision the
                          * reference ships (demo.py:18-29 *_fp16.trt; coreEngine.py:168 fp16 engine_dtype) */
#define ADAS_PREC_FP16X3 3 /* split precision: every activation and weight is a pair of halves (hi, lo * 2^11) = 22 significant
                            * bits, every product three f16 MFMAs (hi*hi + 2^-11 (hi*lo + lo*hi)), fp32 accumulate: f32-class results
                            * (the discrete decisions of the fp32 oracle chain) at a third of the 16-bit MFMA rate instead of
                            * the f32 MFMA's sixteenth */

/* OnnxEngine.__init__(path) / TensorRTEngine.__init__(path) (coreEngine.py:122-126,161-170).
 * `max_batch` frames per call are planned in HBM (the reference is fixed at 1). */
int adas_engine_create(const char* model_path, int precision, int max_batch, adas_engine** out);
int adas_engine_destroy(adas_engine* e);
/* get_engine_input_shape() -> [1,3,H,W] (coreEngine.py:144-145,178-179) */
int adas_engine_input_shape(const adas_engine* e, int64_t dims[4]);
/* get_engine_output_shape() -> (shapes, names) (coreEngine.py:147-148,181-182) */
int adas_engine_num_outputs(const adas_engine* e);
int adas_engine_output_shape(const adas_engine* e, int index, int64_t dims[4], int* ndim);
const char* adas_engine_output_name(const adas_engine* e, int index);
/* engine_inference(input_tensor) (coreEngine.py:150-157,184-186): NCHW fp32 host tensor in, fp32 host
 * tensors out in graph output order.  h_outputs[i] must hold batch * prod(dims[1:]) floats. */
int adas_engine_infer_host(adas_engine* e, const float* h_input_nchw, int batch, float* const* h_outputs);
/* Device-resident form ("frames never round-trip to host"): input is NCHW fp32 already in HBM;
 * outputs stay in HBM and are read through adas_engine_output_device(). Asynchronous on `stream`. */
int adas_engine_infer_device(adas_engine* e, const float* d_input_nchw, int batch, void* stream);
/* 1 when the engine's first layer is the fused stem (bf16 mode) and so can read the packed tensor of adas_preprocess_*_packed */
int adas_engine_accepts_packed_input(const adas_engine* e);
/* ADAS_PREC_* the engine was created with */
int adas_engine_precision(const adas_engine* e);
/* 1 when the model file declared float16 graph inputs (an fp16 ONNX export): the reference's OnnxEngine then reports
 * engine_dtype float16 and exchanges float16 arrays (coreEngine.py:168); the C seam stays fp32 either way. */
int adas_engine_model_io_half(const adas_engine* e);
int adas_engine_infer_device_packed(adas_engine* e, const uint16_t* d_input_nhwc4, int batch, void* stream);
const float* adas_engine_output_device(const adas_engine* e, int index);
/* Algorithmic work of one frame: 2*MACs over conv+linear layers and weight bytes. */
int adas_engine_stats(const adas_engine* e, double* flops_per_frame, double* weight_bytes, int* num_layers);
/* Per-layer device timing of the last adas_engine_profile() call (hipEvents on the engine stream). */
int adas_engine_profile(adas_engine* e, const float* d_input_nchw, int batch, int iters, float* ms_per_layer,
                        int max_layers, int* num_layers);
/* Pipeline-internal fast path of a v8-layout detector whose Detect head runs as the fused kernel (16-bit precisions): while a sink is
 * set, inference writes per anchor the best class probability and its first arg-max class ([batch][num_anchors] each) plus the four
 * box rows of the head, and NOT the head's class rows -- exactly what adas_yolo_post_run derives from them (yoloDetector.py:120-127).
 * Pass the arrays of adas_yolo_post_scan_views and run adas_yolo_post_run_prescanned afterwards; pass NULL, NULL to restore the full
 * head.  adas_pipeline_* does this around its own detector launches; engine_inference callers never see it. */
int adas_engine_detect_sink_supported(const adas_engine* e);
/* Layout (ADAS_HEAD_V8 | ADAS_HEAD_V5), anchor and class count of the head the sink describes: the arrays handed to
 * adas_engine_set_detect_sink hold [max_batch][num_anchors] entries. */
int adas_engine_detect_sink_shape(const adas_engine* e, int32_t* layout, int32_t* num_anchors, int32_t* num_classes);
int adas_engine_set_detect_sink(adas_engine* e, float* d_best_conf, int32_t* d_best_cls);
int adas_engine_layer_info(const adas_engine* e, int layer, char* name, int name_cap, double* flops, int* kind);
/* Which kernel instantiation layer `layer` launches at `batch` frames (matches the rocprofv3 kernel name). */
int adas_engine_layer_kernel(const adas_engine* e, int layer, int batch, char* name, int name_cap);
/* Multi-layer launches (csrc/conv_ml.hip, round 5; OPT-IN: engines created with ADAS_ML=1 in the environment -- at 64 frames the
 * per-layer launches measured faster): at a given batch size, maximal runs of consecutive convolution layers that the
 * per-layer kernels conv_halo / conv_pw would take (the 40x40 / 20x20 layers of the YOLO graphs, the Detect branches) run as ONE
 * persistent launch each -- a table of (layer, tile, channel block) items behind per-layer, per-frame arrival counters -- with results
 * bit-identical to the per-layer launches.  adas_engine_prepare builds the device tables of a batch size
 * (adas_engine_infer_* and adas_pipeline_* call it themselves outside stream captures); adas_engine_ml_info reports what it decided;
 * adas_engine_ml_status synchronises and returns ADAS_ERR_HIP when a dependency wait of the last launch timed out (every wait is
 * bounded: a launch can fail, never hang); adas_engine_launch_count = kernel launches of one forward at that batch. */
int adas_engine_prepare(adas_engine* e, int batch);
int adas_engine_ml_info(const adas_engine* e, int batch, int32_t* n_launches, int32_t* n_layers, int32_t* n_items);
int adas_engine_ml_status(const adas_engine* e, int batch, uint32_t* error_word);
int adas_engine_launch_count(adas_engine* e, int batch);
/* first 16 words of launch `launch`'s control block after its last run: ticket, error word, and (builds with -DADAS_ML_PROF only) the
 * per-phase cycle counters of the item loop (tools/ml_debug.py prof) */
int adas_engine_ml_counters(const adas_engine* e, int batch, int launch, uint32_t head16[16]);
/* The planner of those launches on descriptions alone (no device: tests/test_ml_plan.py).  A layer: kernel (1 = 3x3 halo conv, 4 = 1x1
 * pointwise: CONV_HALO / CONV_PW), stride, activation, residual mode, and its views as (buffer id, pixel stride, channel offset,
 * channels, h, w); `up_c` > 0: the first up_c input channels are read from the half-resolution view `up`.  Outputs: per layer its
 * producer layers after transitive reduction (deps[layer * 6 + k], -1 padded) and the arrivals per frame that complete them
 * (targets[...]); the item table in ticket order (low word: tile or chunk, high word: layer | channel block << 8 | frame << 16);
 * summary = {items, grid, LDS bytes, order mode}.  Returns ADAS_OK or ADAS_ERR_INVALID with the reason in adas_last_error(). */
typedef struct adas_ml_view {
    uint64_t buf;
    int32_t cs, coff, c, h, w;
} adas_ml_view;
typedef struct adas_ml_layer_desc {
    int32_t kernel, stride, act, res_mode, up_c, halo_bn;
    adas_ml_view x, y, res, up;   /* input, output, residual, half-resolution source */
} adas_ml_layer_desc;
int adas_debug_ml_plan(const adas_ml_layer_desc* layers, int n_layers, int batch, int precision, int32_t* deps, int32_t* targets,
                       uint64_t* items, int items_cap, int32_t summary[4]);
/* Debug/parity tap: copy an intermediate activation (by layer index) to the host as NCHW fp32. */
int adas_engine_fetch_activation(adas_engine* e, int layer, int batch, float* h_out_nchw, int64_t dims[4]);

/* ===================================================================================
 * Frame pre-processing on the device (the layer in front of the engine seam): `n` BGR u8 frames
 * (H x W x 3, tightly packed, back to back) in HBM -> the (n,3,h,w) fp32 tensor engine_inference takes.
 * adas_preprocess_yolo replaces YoloDetector.__prepare_input (yoloDetector.py:96-102) =
 * Scaler.process_image (utils.py:42-63) + cv2.dnn.blobFromImage(1/255, swapRB);
 * adas_preprocess_ufld replaces UltrafastLaneDetectorV2.__prepare_input (ultrafastLaneDetectorV2.py:96-112).
 * cv2.resize INTER_LINEAR is restated from OpenCV's 8-bit fixed-point path (parity with cv2 unpinned).
 * =================================================================================== */
int adas_preprocess_yolo(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, float* d_out_nchw, int dst_h,
                         int dst_w, int keep_ratio, void* stream);
int adas_preprocess_ufld(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, float* d_out_nchw, int in_h,
                         int in_w, double crop_ratio, void* stream);
/* The same two conversions into the layout the fused first layer stages anyway: (c0, c1, c2, 0) bf16 pixels, NHWC, 8 bytes per
 * pixel, rounded exactly as the engine rounds the fp32 tensor -- for adas_engine_infer_device_packed (device-resident path:
 * the tensor between pre-processing and the first conv shrinks from 12 to 8 bytes per pixel, same network values). */
int adas_preprocess_yolo_packed(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, uint16_t* d_out_nhwc4, int dst_h,
                                int dst_w, int keep_ratio, void* stream);
int adas_preprocess_ufld_packed(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, uint16_t* d_out_nhwc4, int in_h,
                                int in_w, double crop_ratio, void* stream);
/* The packed form in the 16-bit type of the engine that will consume it: precision = ADAS_PREC_BF16 (what the two calls above
 * write) or ADAS_PREC_FP16. */
int adas_preprocess_yolo_packed_prec(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, uint16_t* d_out_nhwc4, int dst_h,
                                     int dst_w, int keep_ratio, int precision, void* stream);
int adas_preprocess_ufld_packed_prec(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, uint16_t* d_out_nhwc4, int in_h,
                                     int in_w, double crop_ratio, int precision, void* stream);

/* ===================================================================================
 * YOLO post-processing: replaces YoloDetector.__process_output (yoloDetector.py:104-133),
 * Scaler.convert_boxes_coordinate (utils.py:70-87), NMS.fast_soft_nms / NMS.fast_nms
 * (utils.py:161-256 / 105-159), get_nms_results + RectInfo.tolist (yoloDetector.py:135-157,
 * core.py:18-23).  fp64 from the box corners onward; survivor indices bit-exact.
 * =================================================================================== */
typedef struct adas_yolo_post adas_yolo_post;

#define ADAS_HEAD_V8 0 /* (4+nc, A) channel-major: YOLOv8/9/10 (yoloDetector.py:114-115,121-122) */
#define ADAS_HEAD_V5 1 /* (A, 5+nc) row-major, conf = cls*obj in fp32: YOLOv5/6/7 (yoloDetector.py:123-124) */
#define ADAS_HEAD_V5_LITE 2 /* V5 layout holding raw grid offsets: adds YoloLiteParameters.lite_postprocess (yoloDetector.py:35-49) */
#define ADAS_NMS_REFERENCE 0 /* production call yoloDetector.py:139, bug-compatible (SURVEY finding 1) */
#define ADAS_NMS_GREEDY 1    /* NMS.fast_nms, the commented alternative yoloDetector.py:138 */

typedef struct {
    int32_t layout;         /* ADAS_HEAD_V8 | ADAS_HEAD_V5 | ADAS_HEAD_V5_LITE */
    int32_t num_anchors;    /* 8400 | 25200 */
    int32_t num_classes;    /* 80 */
    int32_t nms_mode;       /* ADAS_NMS_* */
    double box_score;       /* keep if conf > box_score (strict), >= 0 */
    double iou_thr;         /* box_nms_iou */
    int32_t pad_h, pad_w;   /* Scaler._pad_shape (utils.py:62) */
    double ratio_h, ratio_w; /* Scaler.get_scale_ratio() (utils.py:65-68) */
    int32_t max_candidates; /* capacity per frame; exceeding it sets ADAS_ERR_CAPACITY on fetch */
    int32_t reserved;
} adas_yolo_post_params;

/* Scaler.process_image geometry without the pixels (utils.py:42-63): fills pad_* and ratio_*. */
int adas_letterbox_params(int src_h, int src_w, int dst_h, int dst_w, int keep_ratio, adas_yolo_post_params* p);

int adas_yolo_post_create(const adas_yolo_post_params* p, int max_batch, adas_yolo_post** out);
int adas_yolo_post_destroy(adas_yolo_post* h);
/* Network input size (YoloLiteParameters.input_shape, yoloDetector.py:30): the v5-lite grid decode derives its three
 * grids from it; required before the first run of an ADAS_HEAD_V5_LITE handle, ignored by the other layouts. */
int adas_yolo_post_set_input_size(adas_yolo_post* h, int in_h, int in_w);
/* d_head: batch head tensors back to back in the reference layout, fp32, in HBM. Asynchronous. */
int adas_yolo_post_run(adas_yolo_post* h, const float* d_head, int batch, void* stream);
/* The same two launches `iters` times on the null stream with events between them: ms[0] = the head scan (per-anchor best class,
 * HBM-bound: the whole head tensor is read once), ms[1] = candidate compaction + inverse letterbox + NMS + RectInfo (latency-bound),
 * averaged per run.  Measurement only (bench.py `post_hbm`). */
/* The post-processing without its class scan: per-anchor (best probability, class) were written into the arrays of
 * adas_yolo_post_scan_views by the producer of the head (adas_engine_set_detect_sink); d_head supplies the box rows. */
int adas_yolo_post_scan_views(adas_yolo_post* h, float** d_best_conf, int32_t** d_best_cls);
int adas_yolo_post_run_prescanned(adas_yolo_post* h, const float* d_head, int batch, void* stream);
int adas_yolo_post_profile(adas_yolo_post* h, const float* d_head, int batch, int iters, float ms[2]);

typedef struct {
    int32_t n_found;       /* anchors over threshold (may exceed capacity) */
    int32_t n_candidates;  /* stored */
    int32_t n_keep;        /* survivors */
    int32_t flags;         /* bit0: capacity overflow */
} adas_yolo_counts;
/* Synchronises, then copies frame `frame`'s results.  Any pointer may be NULL.  Arrays must hold
 * max_candidates entries (x4 for boxes).  cand_*: thresholded rows in anchor order after the inverse
 * letterbox (xywh fp64); keep: NMS result as indices into the candidates, in the reference's order;
 * det_*: RectInfo fields of the survivors; det_xyxy_int = RectInfo.tolist(). */
int adas_yolo_post_fetch(adas_yolo_post* h, int frame, adas_yolo_counts* counts, int32_t* cand_anchor,
                         double* cand_xywh, double* cand_conf, int32_t* cand_cls, int32_t* keep, double* det_xywh,
                         double* det_conf, int32_t* det_cls, int32_t* det_xyxy_int);
/* The survivors only (what yoloDetector.py:141-157 turns into RectInfo), as ONE device-to-host message: a pack kernel behind the NMS
 * writes [counts][n_keep x 64-byte record] and one copy brings it over (a second one past 62 survivors).  Same values, same error
 * behaviour as adas_yolo_post_fetch's keep / det_* arrays; arrays hold max_candidates entries, any pointer may be NULL. */
int adas_yolo_post_fetch_dets(adas_yolo_post* h, int frame, adas_yolo_counts* counts, int32_t* keep, double* det_xywh,
                              double* det_conf, int32_t* det_cls, int32_t* det_xyxy_int);
/* Device views of the survivors for GPU-resident consumers (the tracker): per-frame strides are
 * max_candidates entries.  xyxy as fp64 of the int-truncated corners, scores fp64, classes, counts[4]. */
int adas_yolo_post_device_views(adas_yolo_post* h, const double** d_xyxy, const double** d_score,
                                const int32_t** d_cls, const int32_t** d_counts);
int adas_yolo_post_capacity(const adas_yolo_post* h, int* max_candidates);
/* The head tensor this handle reads: layout (ADAS_HEAD_*), anchors A, classes nc -- (1, 4+nc, A) for V8, (1, A, 5+nc) for V5 /
 * V5_LITE (yoloDetector.py:110-124).  adas_pipeline_create checks the detector engine's output against it. */
int adas_yolo_post_head_shape(const adas_yolo_post* h, int32_t* layout, int32_t* num_anchors, int32_t* num_classes);

/* ===================================================================================
 * EfficientDet post-processing: replaces EfficientdetDetector.__process_output (efficientdetDetector.py:67-85) with
 * Scaler.convert_boxes_coordinate (utils.py:70-87).  Inputs are the exported graph's three outputs (decode and NMS live inside
 * it): boxes (n, 4) xyxy float32 in input pixels, class ids (n) int32, confidences (n) float32.  float32 arithmetic, as in the
 * reference; order kept; a detection is dropped iff conf < box_score.
 * =================================================================================== */
typedef struct adas_effdet_post adas_effdet_post;
typedef struct {
    double box_score;        /* 0.6 in the reference's defaults */
    int32_t pad_h, pad_w;    /* Scaler._pad_shape (fill with adas_letterbox_params) */
    double ratio_h, ratio_w; /* Scaler.get_scale_ratio() */
    int32_t max_boxes;       /* per frame, <= 4096 */
    int32_t reserved;
} adas_effdet_post_params;
int adas_effdet_post_create(const adas_effdet_post_params* p, int max_batch, adas_effdet_post** out);
int adas_effdet_post_destroy(adas_effdet_post* h);
/* Frame b reads h_counts[b] detections at row offset b * max_boxes of the three device arrays. */
int adas_effdet_post_run(adas_effdet_post* h, const float* d_boxes, const int32_t* d_ids, const float* d_confs,
                         const int32_t* h_counts, int batch, void* stream);
/* Synchronises.  xywh [k][4] float32 (RectInfo x, y, width, height), conf [k], class_id [k], xyxy_int [k][4]; returns k in *n_keep. */
int adas_effdet_post_fetch(adas_effdet_post* h, int frame, int32_t* n_keep, float* xywh, float* conf, int32_t* class_id,
                           int32_t* xyxy_int);
/* The in-graph tail of the exported EfficientDet-D0 the reference loads (efficientdetDetector.py:38 OnnxEngine(model_path); its three
 * outputs are read at :68-70): anchor decode, score threshold and per-class NMS over the network's raw head tensors (per pyramid
 * level l = 0..4, stride 8 << l: box regression [batch][cells_l * 9][4] as (dy, dx, dh, dw) and class logits
 * [batch][cells_l * 9][num_classes], rows ordered (y, x, anchor) -- the ten outputs of the "efficientdet-d0" engine graph).  The
 * reference holds no such graph or weights: the steps restate the published post-processing of the architecture (csrc/post_core.h
 * effdet_tail_frame).  Output per frame: boxes (n, 4) xyxy float32 in input pixels, class ids, confidences, by descending
 * confidence -- what adas_effdet_post_run consumes. */
typedef struct adas_effdet_tail adas_effdet_tail;
typedef struct {
    int32_t in_h, in_w;          /* network input (multiples of 128) */
    int32_t num_classes;         /* 90 for the COCO export */
    int32_t max_candidates;      /* anchors over score_thr per frame, <= 3072 (more: ADAS_ERR_CAPACITY at fetch) */
    int32_t max_det;             /* survivors per frame */
    int32_t reserved;
    double score_thr, iou_thr;   /* score > score_thr is a candidate; IoU > iou_thr suppresses within a class */
    double anchor_scale;         /* 4.0 */
} adas_effdet_tail_params;
int adas_effdet_tail_create(const adas_effdet_tail_params* p, int max_batch, adas_effdet_tail** out);
int adas_effdet_tail_destroy(adas_effdet_tail* h);
int adas_effdet_tail_run(adas_effdet_tail* h, const float* const* d_reg /* [5] */, const float* const* d_cls /* [5] */, int batch, void* stream);
/* Synchronises.  boxes_xyxy [n][4], class_id [n], conf [n]; n_candidates (optional) = anchors over the threshold. */
int adas_effdet_tail_fetch(adas_effdet_tail* h, int frame, int32_t* n_det, float* boxes_xyxy, int32_t* class_id, float* conf,
                           int32_t* n_candidates);
/* Device-resident results: boxes [max_batch][max_det][4], ids / confs [max_batch][max_det], counts [max_batch][2] (survivors, candidates). */
int adas_effdet_tail_device_views(adas_effdet_tail* h, const float** d_boxes, const int32_t** d_ids, const float** d_confs,
                                  const int32_t** d_counts);
/* EfficientdetDetector.__prepare_input (efficientdetDetector.py:57-65): Scaler.process_image letterbox (canvas 114), then
 * (pixel / 255 - mean) / std per BGR channel -- NO channel swap -- with mean (0.406, 0.456, 0.485), std (0.225, 0.224, 0.229),
 * evaluated in double and cast to float32; NCHW. */
int adas_preprocess_effdet(const uint8_t* d_frames_bgr, int n, int src_h, int src_w, float* d_out_nchw, int dst_h, int dst_w,
                           int keep_ratio, void* stream);

/* ===================================================================================
 * UFLDv2 lane decode: replaces UltrafastLaneDetectorV2.__process_output
 * (ultrafastLaneDetectorV2.py:114-181, _softmax :15-19, ModelConfig :21-55)
 * =================================================================================== */
typedef struct adas_ufld_decode adas_ufld_decode;
#define ADAS_UFLD_MAX_POINTS 128

typedef struct {
    int32_t grid_row, cls_row, grid_col, cls_col; /* 200,72,100,81 (CULane) */
    int32_t img_w, img_h;                         /* source image size the points are scaled to */
    int32_t local_width;                          /* 1 */
    int32_t num_lanes;                            /* last tensor dimension: 4 (CULane / Tusimple), 10 (CurveLanes configs,
                                                   * configs/curvelanes_res18.py:25); 0 = 4.  Lanes 1,2 (rows) and 0,3 (columns)
                                                   * are decoded whatever the count (ultrafastLaneDetectorV2.py:141-142) */
    const double* h_row_anchor;                   /* [cls_row] cfg.row_anchor (host) */
    const double* h_col_anchor;                   /* [cls_col] cfg.col_anchor (host) */
} adas_ufld_params;

int adas_ufld_decode_create(const adas_ufld_params* p, int max_batch, adas_ufld_decode** out);
int adas_ufld_decode_destroy(adas_ufld_decode* h);
/* Four device tensors in the engine's output layout (1,G,K,4); frame b is at ptr + b*batch_stride_* */
int adas_ufld_decode_run(adas_ufld_decode* h, const float* d_loc_row, const float* d_loc_col,
                         const float* d_exist_row, const float* d_exist_col, size_t stride_loc_row,
                         size_t stride_loc_col, size_t stride_exist_row, size_t stride_exist_col, int batch,
                         void* stream);
/* lane order: left-side, left-ego, right-ego, right-side (ultrafastLaneDetectorV2.py:143-145).
 * points: [4][ADAS_UFLD_MAX_POINTS][2] (x,y) int32; counts[4]; detected[4]. */
int adas_ufld_decode_fetch(adas_ufld_decode* h, int frame, int32_t* points, int32_t* counts, int32_t* detected);
/* The inverse of fetch: places lane points decoded elsewhere into frame `frame`'s slot (same array shapes), e.g. to run
 * adas_lane_geometry on them. */
int adas_ufld_decode_upload(adas_ufld_decode* h, int frame, const int32_t* points, const int32_t* counts, const int32_t* detected);

/* -----------------------------------------------------------------------------------
 * UFLD (v1) lane decode: replaces UltrafastLaneDetector.__process_output (ultrafastLaneDetector.py:96-139,
 * ModelConfig :16-40).  One (1, griding_num+1, cls_num_per_lane, 4) tensor per frame; the handle type, fetch and
 * destroy are shared with the v2 decoder (lane index = the tensor's lane axis, points in the reference's order).
 * ----------------------------------------------------------------------------------- */
typedef struct {
    int32_t griding_num, cls_num_per_lane; /* 100,56 (Tusimple) | 200,18 (CULane) */
    int32_t cfg_img_w, cfg_img_h;          /* ModelConfig.img_w/img_h: 1280x720 | 1640x590 */
    int32_t input_w, input_h;              /* network input, 800x288 */
    int32_t src_w, src_h;                  /* source frame size: w_ratio/h_ratio of ultrafastLaneDetector.py:80 */
    const double* h_row_anchor;            /* [cls_num_per_lane] cfg.row_anchor (host) */
} adas_ufld1_params;
int adas_ufld1_decode_create(const adas_ufld1_params* p, int max_batch, adas_ufld_decode** out);
int adas_ufld1_decode_set_source_size(adas_ufld_decode* h, int src_w, int src_h);
/* 1: handle made by adas_ufld1_decode_create (UFLD v1), 2: by adas_ufld_decode_create (UFLDv2), 0: NULL */
int adas_ufld_decode_kind(const adas_ufld_decode* h);
/* The tensors this handle decodes, in engine output order: returns their number (4 for UFLDv2: loc_row, loc_col, exist_row,
 * exist_col, ultrafastLaneDetectorV2.py:118; 1 for UFLD v1) and fills dims[i] = (1, G, K, 4).  < 0: error. */
int adas_ufld_decode_expected_outputs(const adas_ufld_decode* h, int64_t dims[4][4]);
int adas_ufld1_decode_run(adas_ufld_decode* h, const float* d_out, size_t batch_stride, int batch, void* stream);

/* -----------------------------------------------------------------------------------
 * Ego-lane geometry on the decoder's device-resident points: replaces
 * LaneDetectBase.__update_lanes_status / __update_lanes_area / __adjust_lanes_points (ufldDetector/core.py:102-158),
 * PerspectiveTransformation.transformToBirdViewPoints and .calcCurveAndOffset (perspectiveTransformation.py:120-214,
 * without the drawing calls).  The homography itself stays a host decision (updateTransformParams, :39-86).
 * ----------------------------------------------------------------------------------- */
typedef struct adas_lane_geometry adas_lane_geometry;
typedef struct {
    int32_t img_h;         /* source frame height = resampling count of __adjust_lanes_points (core.py:130) */
    int32_t bird_w, bird_h; /* PerspectiveTransformation.img_size (bird-view image); bird_h must exceed 719 for the curvature
                              (the reference reads row 719, perspectiveTransformation.py:196-199) */
    int32_t adjust_lanes;  /* default for run() */
    double M[9];           /* frontal -> bird-view homography, row-major (cv2.getPerspectiveTransform, :33) */
} adas_lane_geometry_params;
typedef struct {
    int32_t area_status;          /* both ego lanes detected (core.py:143-148) */
    int32_t n_area_left, n_area_right; /* area_points = left points then the reversed right points (core.py:158) */
    int32_t direction;            /* 0: no curve estimate, 1 "L", 2 "R", 3 "F" (perspectiveTransformation.py:170-175) */
    int32_t bird_counts[4];       /* points per lane in the bird view */
    double curvature;             /* metres (:193) */
    double offset;                /* metres from the lane centre (:201-202) */
} adas_lane_geometry_result;
int adas_lane_geometry_create(const adas_lane_geometry_params* p, int max_batch, adas_lane_geometry** out);
int adas_lane_geometry_destroy(adas_lane_geometry* h);
int adas_lane_geometry_set_matrix(adas_lane_geometry* h, const double* M9);
/* Reads the lane points the decoder (v1 or v2 handle) left in HBM for frames [0, batch). Asynchronous.
 * adjust_lanes < 0: the value given at create. */
int adas_lane_geometry_run(adas_lane_geometry* h, const adas_ufld_decode* decode, int adjust_lanes, int batch, void* stream);
/* area_points: room for [2*img_h][2] int32 (x,y); bird_points: [4][ADAS_UFLD_MAX_POINTS][2]; either may be NULL. */
int adas_lane_geometry_fetch(adas_lane_geometry* h, int frame, adas_lane_geometry_result* res, int32_t* area_points,
                             int32_t* bird_points);

/* ===================================================================================
 * ByteTrack: replaces BYTETracker.__init__/update/reset (byteTracker.py:30-51,62-185,187-200)
 * with matching.py, kalman_filter.py, strack.py, base_track.py, byteTrack/utils.py underneath.
 * One independent tracker (own id counter) per stream; the whole update runs on the GPU.
 * =================================================================================== */
typedef struct adas_bytetrack adas_bytetrack;

typedef struct {
    double track_thresh;  /* 0.5 */
    double match_thresh;  /* 0.8 */
    double frame_rate;    /* 30 */
    int32_t track_buffer; /* 30 */
    int32_t max_tracks;   /* track-table capacity per stream (tracked + lost) */
    int32_t max_dets;     /* detections per frame capacity */
    int32_t reserved;
} adas_bytetrack_params;

typedef struct {          /* base_track.py:61-72 + strack.py:207-215 (crops stay on the host; trajectories: adas_bytetrack_fetch_trajectories) */
    double tlwh[4];       /* STrack.tlwh (Kalman-filtered) */
    double score;
    int32_t track_id, state, is_activated, class_id;
    int32_t frame_id, start_frame, tracklet_len;
    int32_t trajectory_len;   /* len(STrack.trajectories), 0 .. 30 (strack.py:53,115); the boxes: adas_bytetrack_fetch_trajectories */
} adas_track;
#define ADAS_TRAJECTORY_LEN 30   /* LimitedList(30), strack.py:53 */

typedef struct {
    int32_t frame_id, id_count, n_tracked, n_lost, err, pad[3];
} adas_track_header;

int adas_bytetrack_create(const adas_bytetrack_params* p, int n_streams, adas_bytetrack** out);
int adas_bytetrack_destroy(adas_bytetrack* h);
int adas_bytetrack_reset(adas_bytetrack* h, int stream_index /* -1 = all */);
/* BYTETracker.update(bboxes, scores, class_ids, frame) for one stream from host arrays
 * (xyxy fp64 [n][4], scores fp64, integer class ids). */
int adas_bytetrack_update_host(adas_bytetrack* h, int stream_index, const double* h_xyxy, const double* h_scores,
                               const int32_t* h_cls, int n);
/* All streams at once from device-resident detections (e.g. adas_yolo_post_device_views):
 * stream s reads n = d_counts[s*count_stride + count_index] rows at row offset s*det_stride. */
int adas_bytetrack_update_device(adas_bytetrack* h, const double* d_xyxy, const double* d_scores,
                                 const int32_t* d_cls, const int32_t* d_counts, int det_stride, int count_stride,
                                 int count_index, int n_streams, void* stream);
/* The same for n_frames CONSECUTIVE frames of every stream in one launch (temporal micro-batching, adas_pipeline_desc.micro_batch):
 * frame f of stream s reads detection slab f * n_streams + s; the updates of a stream run in temporal order inside its workgroup,
 * exactly as n_frames calls of adas_bytetrack_update_device would (BYTETracker.update once per frame, byteTracker.py:62). */
int adas_bytetrack_update_device_frames(adas_bytetrack* h, const double* d_xyxy, const double* d_scores,
                                        const int32_t* d_cls, const int32_t* d_counts, int det_stride, int count_stride,
                                        int count_index, int n_streams, int n_frames, void* stream);
/* Sizes the per-frame message store of adas_bytetrack_update_device_frames ahead of time (needed before such a launch is captured into
 * a hipGraph: adas_pipeline_create does it for micro-batched pipelines; eager launches size it on demand). */
int adas_bytetrack_reserve_frames(adas_bytetrack* h, int n_frames, int n_streams);
/* The message of frame `frame` (0 .. n_frames - 1) of the LAST adas_bytetrack_update_device_frames launch: what BYTETracker.update returned
 * for that frame (byteTracker.py:185) -- the live state (adas_bytetrack_fetch) only holds the last frame's.  Same layout as fetch. */
int adas_bytetrack_fetch_frame(adas_bytetrack* h, int stream_index, int frame, adas_track_header* hdr, adas_track* tracks, int max_tracks);
/* Synchronises; tracks[0..n_tracked) are tracked_stracks, then n_lost lost_stracks, list order kept. */
int adas_bytetrack_fetch(adas_bytetrack* h, int stream_index, adas_track_header* hdr, adas_track* tracks,
                         int max_tracks);

/* STrack.trajectories of the LIVE state (the last 30 detection boxes a track was updated with, strack.py:115 -- what DrawTrackedOnFrame
 * draws and filter_trajectories / plot_directions read, byteTracker.py:202-215), for the tracks in the order of adas_bytetrack_fetch
 * (tracked, then lost): lens[k] = len(trajectories) and tlbr[k][i][0..4) = trajectories[i] (x1, y1, x2, y2 fp64, oldest first) for
 * i < lens[k].  lens holds max_tracks entries, tlbr max_tracks x ADAS_TRAJECTORY_LEN x 4; *n_tracks = n_tracked + n_lost.  Synchronises. */
int adas_bytetrack_fetch_trajectories(adas_bytetrack* h, int stream_index, int32_t* lens, double* tlbr, int max_tracks, int32_t* n_tracks);

/* ===================================================================================
 * Fused per-frame pipeline (the path demo.py:261-281 drives): detector forward + decode/NMS,
 * lane forward + decode, tracker update, for `n_streams` independent video streams per step,
 * captured in a hipGraph.  Inputs are the two pre-processed NCHW fp32 tensors per stream in HBM.
 * =================================================================================== */
typedef struct adas_pipeline adas_pipeline;
typedef struct {
    adas_engine* detector;      /* may be NULL */
    adas_engine* lane;          /* may be NULL */
    adas_yolo_post* post;       /* required with detector */
    adas_ufld_decode* decode;   /* required with lane */
    adas_bytetrack* tracker;    /* may be NULL */
    int32_t n_streams;
    int32_t use_graph;          /* bit0: capture the step in a hipGraph (the lane branch is then forked onto a second
                                 * stream so the two nets overlap); bit1: keep both nets on one stream */
    adas_lane_geometry* geometry; /* may be NULL; with lane: area polygon / bird view / curvature right behind the decode */
    int32_t micro_batch;        /* 0 / 1: one frame of every stream per step.  B > 1: temporal micro-batching -- a step takes B
                                 * CONSECUTIVE frames of every stream (frame b of stream s at index b * n_streams + s of the input),
                                 * the pre-processing / networks / decode / NMS run on all n_streams * B frames at once (no
                                 * cross-frame dependency there) and the tracker then consumes each stream's B frames in order
                                 * (B update launches).  Engines, post and decode handles need max_batch >= n_streams * B; results of
                                 * frame b of stream s are fetched at frame index b * n_streams + s, tracker state per stream.
                                 * Throughput mode for few streams per GPU (adds B - 1 frames of latency). */
    int32_t reserved;
} adas_pipeline_desc;
int adas_pipeline_create(const adas_pipeline_desc* d, adas_pipeline** out);
/* 1 when the steps feed the post-processing's per-anchor scan arrays from the fused Detect kernel (adas_engine_set_detect_sink). */
int adas_pipeline_detect_sink(const adas_pipeline* p);
int adas_pipeline_destroy(adas_pipeline* p);
/* One step = one frame of every stream (micro_batch frames with temporal micro-batching).  Asynchronous; adas_pipeline_sync() waits. */
int adas_pipeline_step(adas_pipeline* p, const float* d_det_input_nchw, const float* d_lane_input_nchw);
/* The same step from camera frames: n_streams BGR u8 frames (src_h x src_w x 3, back to back) in HBM; each branch runs its
 * pre-processing (adas_preprocess_yolo / adas_preprocess_ufld with lane_crop_ratio = ModelConfig.crop_ratio) into seam
 * tensors the pipeline owns, so one upload per stream and frame feeds both nets (yoloDetector.py:96-102,
 * ultrafastLaneDetectorV2.py:96-112 + the body of demo.py:261-281). */
int adas_pipeline_step_frames(adas_pipeline* p, const uint8_t* d_frames_bgr, int src_h, int src_w, double lane_crop_ratio);
/* The same step from HOST frames (what a capture thread hands over, demo.py:261-270): the frames of this step are copied to one
 * of two device staging buffers on a dedicated copy stream while the previous step still computes (double buffering: copy k+1
 * overlaps compute k; the compute stream waits for its copy, the copy stream for the step that last read its buffer).
 * h_frames_bgr should come from adas_host_alloc (pinned): pageable memory works but the copy then cannot overlap. */
int adas_pipeline_step_frames_host(adas_pipeline* p, const uint8_t* h_frames_bgr, int src_h, int src_w, double lane_crop_ratio);
/* The call above returns when the upload is ENQUEUED: the host buffer must not be rewritten before the copy has read it.  This waits
 * for exactly that (the copy stream only, not the compute): call it before refilling a host buffer that the last step was given. */
int adas_pipeline_wait_upload(adas_pipeline* p);
int adas_pipeline_sync(adas_pipeline* p);
/* Device time of the last `n` steps' sections in ms (hipEvents on the pipeline stream):
 * [0] detector net, [1] yolo post, [2] lane net, [3] lane decode, [4] tracker, [5] whole step. */
int adas_pipeline_timings(adas_pipeline* p, float ms[6]);

#ifdef __cplusplus
}
#endif
#endif /* ADAS_HIP_H */
