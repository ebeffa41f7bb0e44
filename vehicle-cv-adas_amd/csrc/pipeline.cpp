// pipeline.cpp -- the fused per-frame ADAS step (detector + NMS, lane + decode, tracker) for
// n_streams independent video streams, optionally replayed from a hipGraph.
// Mirrors what demo.py:261-281 drives per frame, minus UI; nothing returns to the host per step.
#include "engine.h"
#include <stdlib.h>

struct adas_yolo_post;
struct adas_ufld_decode;
struct adas_bytetrack;

struct adas_pipeline {
    adas_pipeline_desc d;
    hipStream_t st = nullptr;
    hipStream_t st_lane = nullptr;  // graph mode: the lane branch is captured on its own stream so the two nets overlap
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // A captured step bakes every kernel argument in: the input pointers, the source geometry, the lane crop ratio and the
    // post / decode / geometry configuration (the latter through adas::config_generation()).  All of them are the cache key.
    struct GraphKey {
        const void* in_a;   // detector seam tensor, or the u8 frames
        const void* in_b;   // lane seam tensor (null for a step_frames entry)
        int src_h, src_w;   // step_frames: camera geometry (0 for a seam-tensor entry)
        double crop;        // step_frames: lane crop ratio
        bool operator==(const GraphKey& o) const {
            return in_a == o.in_a && in_b == o.in_b && src_h == o.src_h && src_w == o.src_w && crop == o.crop;
        }
    };
    struct Cached {
        GraphKey key;
        hipGraph_t graph;
        hipGraphExec_t exec;
    };
    std::vector<Cached> graphs;              // one captured step per distinct key
    unsigned long long graphs_gen = 0;       // config generation the cached captures were recorded under
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool timed = false;
    bool sink = false;   // the detector's fused v8 Detect feeds the post-processing's scan arrays directly (decided at create)
    // step_frames: u8 camera frames in, the engine-seam tensors live here
    float* det_in = nullptr;
    float* lane_in = nullptr;
    struct FrameSrc {
        const uint8_t* frames;
        int h, w;
        double crop;
    };
    FrameSrc fsrc{nullptr, 0, 0, 0.0};  // set while a step_frames call records
    bool packed = false;  // both first layers are the fused stem: the seam tensors are (c0,c1,c2,0) 16-bit NHWC, 8 B per pixel
    // step_frames_host: two device frame buffers filled by a copy stream
    hipStream_t st_copy = nullptr;
    uint8_t* host_stage[2] = {nullptr, nullptr};
    size_t host_stage_bytes = 0;
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
    unsigned long long host_steps = 0;
};

using namespace adas;

// events == true: everything on one stream with section events (per-stage timing).
// events == false: the lane branch (net + decode) runs on st_lane, forked/joined with events, so the
// latency-bound detector layers and the MFMA-bound lane layers share the chip (independent work:
// demo.py:261-281 runs them back to back only because the reference is single-threaded Python).
static int record_step(adas_pipeline* p, const float* d_det, const float* d_lane, bool events) {
    const int NS = p->d.n_streams;                                   // streams (tracker instances)
    const int B = p->d.micro_batch > 1 ? p->d.micro_batch : 1;       // consecutive frames of each stream in this step
    const int S = NS * B;                                            // frames through pre-processing, nets, decode and NMS
    hipStream_t st = p->st;
    const bool fork = !events && p->d.detector && p->d.lane && !(p->d.use_graph & 2);
    hipStream_t sl = fork ? p->st_lane : st;
    int rc;
    if (events) ADAS_HIP_TRY(hipEventRecord(p->ev[0], st));
    if (fork) {
        ADAS_HIP_TRY(hipEventRecord(p->ev_fork, st));
        ADAS_HIP_TRY(hipStreamWaitEvent(sl, p->ev_fork, 0));
    }
    if (p->fsrc.frames) {  // pre-processing inside the step: each branch converts the shared u8 frames for its own net
        if (p->d.detector) {
            int64_t is[4];
            adas_engine_input_shape(p->d.detector, is);
            rc = p->packed ? adas_preprocess_yolo_packed_prec(p->fsrc.frames, S, p->fsrc.h, p->fsrc.w, (uint16_t*)p->det_in, (int)is[2], (int)is[3], 1,
                                                              p->d.detector->prec, st)
                           : adas_preprocess_yolo(p->fsrc.frames, S, p->fsrc.h, p->fsrc.w, p->det_in, (int)is[2], (int)is[3], 1, st);
            if (rc) return rc;
        }
        if (p->d.lane) {
            int64_t is[4];
            adas_engine_input_shape(p->d.lane, is);
            rc = p->packed ? adas_preprocess_ufld_packed_prec(p->fsrc.frames, S, p->fsrc.h, p->fsrc.w, (uint16_t*)p->lane_in, (int)is[2], (int)is[3], p->fsrc.crop,
                                                              p->d.lane->prec, sl)
                           : adas_preprocess_ufld(p->fsrc.frames, S, p->fsrc.h, p->fsrc.w, p->lane_in, (int)is[2], (int)is[3], p->fsrc.crop, sl);
            if (rc) return rc;
        }
    }
    const bool packed_now = p->fsrc.frames && p->packed;
    if (p->d.detector) {
        // fused v8 Detect: per-anchor (best probability, class) come straight out of its registers into the post-processing's scan arrays;
        // the head's class rows are not written and the scan launch goes away (ADAS_NO_DETECT_SINK=1: the full head + scan)
        float* sc_conf = nullptr;
        int32_t* sc_cls = nullptr;
        if (p->sink) {
            rc = adas_yolo_post_scan_views(p->d.post, &sc_conf, &sc_cls);
            if (rc) return rc;
            rc = adas_engine_set_detect_sink(p->d.detector, sc_conf, sc_cls);
            if (rc) return rc;
        }
        rc = packed_now ? adas_engine_infer_device_packed(p->d.detector, (const uint16_t*)d_det, S, st) : adas_engine_infer_device(p->d.detector, d_det, S, st);
        if (p->sink) (void)adas_engine_set_detect_sink(p->d.detector, nullptr, nullptr);   // launches are enqueued (or captured) with the sink baked in
        if (rc) return rc;
        if (events) ADAS_HIP_TRY(hipEventRecord(p->ev[1], st));
        rc = p->sink ? adas_yolo_post_run_prescanned(p->d.post, adas_engine_output_device(p->d.detector, 0), S, st)
                     : adas_yolo_post_run(p->d.post, adas_engine_output_device(p->d.detector, 0), S, st);
        if (rc) return rc;
    } else if (events) ADAS_HIP_TRY(hipEventRecord(p->ev[1], st));
    if (events) ADAS_HIP_TRY(hipEventRecord(p->ev[2], st));
    if (p->d.lane) {
        rc = packed_now ? adas_engine_infer_device_packed(p->d.lane, (const uint16_t*)d_lane, S, sl) : adas_engine_infer_device(p->d.lane, d_lane, S, sl);
        if (rc) return rc;
        if (events) ADAS_HIP_TRY(hipEventRecord(p->ev[3], st));
        const adas_engine* le = p->d.lane;
        size_t stride = (size_t)le->bufs[le->outs[0].buf].h * le->bufs[le->outs[0].buf].w * le->bufs[le->outs[0].buf].c;
        if (adas_ufld_decode_kind(p->d.decode) == 1)  // UFLD v1: one (G+1, K, 4) tensor per frame
            rc = adas_ufld1_decode_run(p->d.decode, adas_engine_output_device(le, 0), stride, S, sl);
        else
            rc = adas_ufld_decode_run(p->d.decode, adas_engine_output_device(le, 0), adas_engine_output_device(le, 1),
                                      adas_engine_output_device(le, 2), adas_engine_output_device(le, 3), stride, stride, stride, stride, S, sl);
        if (rc) return rc;
        if (p->d.geometry) {
            rc = adas_lane_geometry_run(p->d.geometry, p->d.decode, -1, S, sl);
            if (rc) return rc;
        }
    } else if (events) ADAS_HIP_TRY(hipEventRecord(p->ev[3], st));
    if (events) ADAS_HIP_TRY(hipEventRecord(p->ev[4], st));
    if (p->d.tracker && p->d.detector) {
        const double *xy, *sc;
        const int32_t *cl, *cn;
        rc = adas_yolo_post_device_views(p->d.post, &xy, &sc, &cl, &cn);
        if (rc) return rc;
        int cap = 0;
        rc = adas_yolo_post_capacity(p->d.post, &cap);
        if (rc) return rc;
        // frame b of stream s sits at frame index b * NS + s: one launch, every stream's workgroup runs its B updates in temporal order
        rc = B > 1 ? adas_bytetrack_update_device_frames(p->d.tracker, xy, sc, cl, cn, cap, 4, 2, NS, B, st)
                   : adas_bytetrack_update_device(p->d.tracker, xy, sc, cl, cn, cap, 4, 2, NS, st);
        if (rc) return rc;
    }
    if (fork) {
        ADAS_HIP_TRY(hipEventRecord(p->ev_join, sl));
        ADAS_HIP_TRY(hipStreamWaitEvent(st, p->ev_join, 0));
    }
    if (events) ADAS_HIP_TRY(hipEventRecord(p->ev[5], st));
    return ADAS_OK;
}

static void drop_graphs(adas_pipeline* p) {
    for (auto& g : p->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    p->graphs.clear();
}

extern "C" {

int adas_pipeline_create(const adas_pipeline_desc* d, adas_pipeline** out) {
    ADAS_REQUIRE(d && out && d->n_streams > 0, ADAS_ERR_INVALID, "adas_pipeline_create: bad argument");
    ADAS_REQUIRE(d->detector || d->lane, ADAS_ERR_INVALID, "pipeline needs a detector and/or a lane engine");
    ADAS_REQUIRE(!d->detector || d->post, ADAS_ERR_INVALID, "detector engine needs a yolo_post handle");
    ADAS_REQUIRE(!d->lane || d->decode, ADAS_ERR_INVALID, "lane engine needs a ufld_decode handle");
    const int lane_outs = d->lane ? (adas_ufld_decode_kind(d->decode) == 1 ? 1 : 4) : 0;  // ultrafastLaneDetector.py:73-75 | V2:93-94
    ADAS_REQUIRE(!d->lane || adas_engine_num_outputs(d->lane) == lane_outs, ADAS_ERR_INVALID,
                 "Output dims is error, please check model. load %d channels not match %d.", d->lane ? adas_engine_num_outputs(d->lane) : 0, lane_outs);
    if (d->detector) {  // the post kernel indexes the head by ITS (layout, A, nc): a mismatch would read past the engine's buffer
        int32_t layout = 0, A = 0, nc = 0;
        int rc0 = adas_yolo_post_head_shape(d->post, &layout, &A, &nc);
        if (rc0) return rc0;
        int64_t od[4] = {0, 0, 0, 0};
        int nd = 0;
        ADAS_REQUIRE(adas_engine_num_outputs(d->detector) >= 1 && adas_engine_output_shape(d->detector, 0, od, &nd) == ADAS_OK && nd == 3,
                     ADAS_ERR_INVALID, "detector engine has no (1, C, A) / (1, A, C) head output");
        const int64_t want1 = layout == ADAS_HEAD_V8 ? 4 + nc : A, want2 = layout == ADAS_HEAD_V8 ? A : 5 + nc;
        ADAS_REQUIRE(od[1] == want1 && od[2] == want2, ADAS_ERR_INVALID,
                     "detector head is (1,%lld,%lld) but the post-processor was created for (1,%lld,%lld) [layout %d, %d anchors, %d classes]",
                     (long long)od[1], (long long)od[2], (long long)want1, (long long)want2, layout, A, nc);
    }
    if (d->lane) {
        int64_t want[4][4];
        const int n = adas_ufld_decode_expected_outputs(d->decode, want);
        for (int i = 0; i < n; ++i) {
            int64_t od[4] = {0, 0, 0, 0};
            int nd = 0;
            ADAS_REQUIRE(adas_engine_output_shape(d->lane, i, od, &nd) == ADAS_OK && nd == 4 && od[1] == want[i][1] && od[2] == want[i][2] && od[3] == want[i][3],
                         ADAS_ERR_INVALID, "lane output %d is (1,%lld,%lld,%lld) but the decoder was created for (1,%lld,%lld,%lld)", i, (long long)od[1],
                         (long long)od[2], (long long)od[3], (long long)want[i][1], (long long)want[i][2], (long long)want[i][3]);
        }
    }
    const int frames_per_step = d->n_streams * (d->micro_batch > 1 ? d->micro_batch : 1);
    ADAS_REQUIRE(d->micro_batch >= 0 && d->micro_batch <= 64, ADAS_ERR_INVALID, "micro_batch must be in [0, 64]");
    ADAS_REQUIRE(!d->detector || frames_per_step <= d->detector->max_batch, ADAS_ERR_INVALID, "n_streams x micro_batch exceeds detector max_batch");
    ADAS_REQUIRE(!d->lane || frames_per_step <= d->lane->max_batch, ADAS_ERR_INVALID, "n_streams x micro_batch exceeds lane max_batch");
    // the post-processing handles index per-frame arenas: too small a handle must fail here, not inside the first (captured) step
    ADAS_REQUIRE(!d->post || frames_per_step <= adas::handle_max_batch(d->post), ADAS_ERR_INVALID,
                 "n_streams x micro_batch = %d exceeds the yolo_post handle's max_batch %d", frames_per_step, adas::handle_max_batch(d->post));
    ADAS_REQUIRE(!d->decode || frames_per_step <= adas::handle_max_batch(d->decode), ADAS_ERR_INVALID,
                 "n_streams x micro_batch = %d exceeds the ufld_decode handle's max_batch %d", frames_per_step, adas::handle_max_batch(d->decode));
    ADAS_REQUIRE(!d->geometry || frames_per_step <= adas::handle_max_batch(d->geometry), ADAS_ERR_INVALID,
                 "n_streams x micro_batch = %d exceeds the lane_geometry handle's max_batch %d", frames_per_step, adas::handle_max_batch(d->geometry));
    adas_pipeline* p = new adas_pipeline();
    p->d = *d;
    if (hipStreamCreateWithFlags(&p->st, hipStreamNonBlocking) != hipSuccess) {
        delete p;
        return hip_fail(hipGetLastError(), "hipStreamCreate", __FILE__, __LINE__);
    }
    if (hipStreamCreateWithFlags(&p->st_lane, hipStreamNonBlocking) != hipSuccess) {
        adas_pipeline_destroy(p);
        return hip_fail(hipGetLastError(), "hipStreamCreate", __FILE__, __LINE__);
    }
    for (auto& e : p->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            adas_pipeline_destroy(p);
            return hip_fail(hipGetLastError(), "hipEventCreate", __FILE__, __LINE__);
        }
    if (hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) != hipSuccess) {
        adas_pipeline_destroy(p);
        return hip_fail(hipGetLastError(), "hipEventCreate", __FILE__, __LINE__);
    }
    if (d->tracker && d->detector && d->micro_batch > 1) {   // every frame's tracker message stays fetchable (adas_bytetrack_fetch_frame)
        int rc = adas_bytetrack_reserve_frames(d->tracker, d->micro_batch, d->n_streams);
        if (rc) {
            adas_pipeline_destroy(p);
            return rc;
        }
    }
    if (d->detector && d->post) {
        // only when the post-processing was created for exactly the head the engine's fused Detect describes (its scan arrays hold
        // [max_batch][num_anchors] entries of that layout)
        int32_t layout = -1, pa = 0, pn = 0, el = -2, ea = 0, en = 0;
        const char* env = getenv("ADAS_NO_DETECT_SINK");
        (void)adas_yolo_post_head_shape(d->post, &layout, &pa, &pn);
        p->sink = adas_engine_detect_sink_supported(d->detector) && adas_engine_detect_sink_shape(d->detector, &el, &ea, &en) == ADAS_OK &&
                  el == layout && ea == pa && en == pn && !(env && env[0] == '1');
    }
    *out = p;
    return ADAS_OK;
}
int adas_pipeline_detect_sink(const adas_pipeline* p) { return p && p->sink ? 1 : 0; }

int adas_pipeline_destroy(adas_pipeline* p) {
    if (!p) return ADAS_OK;
    drop_graphs(p);
    for (auto& e : p->ev)
        if (e) (void)hipEventDestroy(e);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    if (p->st_lane) (void)hipStreamDestroy(p->st_lane);
    if (p->st) (void)hipStreamDestroy(p->st);
    if (p->det_in) (void)hipFree(p->det_in);
    if (p->lane_in) (void)hipFree(p->lane_in);
    for (int k = 0; k < 2; ++k) {
        if (p->host_stage[k]) (void)hipFree(p->host_stage[k]);
        if (p->ev_copied[k]) (void)hipEventDestroy(p->ev_copied[k]);
        if (p->ev_consumed[k]) (void)hipEventDestroy(p->ev_consumed[k]);
    }
    if (p->st_copy) (void)hipStreamDestroy(p->st_copy);
    delete p;
    return ADAS_OK;
}

// Replays the captured step for `key`, capturing it first when no capture of the current configuration exists.
static int replay_step(adas_pipeline* p, const adas_pipeline::GraphKey& key, const float* d_det, const float* d_lane,
                       const adas_pipeline::FrameSrc* fs) {
    const unsigned long long gen = adas::config_generation();
    if (gen != p->graphs_gen) {  // a post / decode / geometry setter ran since the captures were made: their arguments are stale
        ADAS_HIP_TRY(hipStreamSynchronize(p->st));
        drop_graphs(p);
        p->graphs_gen = gen;
    }
    hipGraphExec_t exec = nullptr;
    for (auto& g : p->graphs)
        if (g.key == key) exec = g.exec;
    if (!exec) {
        ADAS_REQUIRE(p->graphs.size() < 64, ADAS_ERR_CAPACITY, "more than 64 distinct input buffers: reuse staging buffers with use_graph");
        adas_pipeline::Cached g{key, nullptr, nullptr};
        if (fs) p->fsrc = *fs;
        {   // device tables of the engines' multi-layer launches are built BEFORE the capture (allocations and copies cannot be captured)
            const int frames = p->d.n_streams * (p->d.micro_batch > 1 ? p->d.micro_batch : 1);
            int prc = p->d.detector ? adas::engine_prepare(p->d.detector, frames) : ADAS_OK;
            if (prc == ADAS_OK && p->d.lane) prc = adas::engine_prepare(p->d.lane, frames);
            if (prc != ADAS_OK) return prc;
        }
        ADAS_HIP_TRY(hipStreamBeginCapture(p->st, hipStreamCaptureModeThreadLocal));
        int rc = record_step(p, d_det, d_lane, false);
        hipError_t ce = hipStreamEndCapture(p->st, &g.graph);
        p->fsrc.frames = nullptr;
        if (rc == ADAS_OK && ce != hipSuccess) rc = hip_fail(ce, "hipStreamEndCapture", __FILE__, __LINE__);
        if (rc == ADAS_OK) {
            hipError_t ie = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
            if (ie != hipSuccess) rc = hip_fail(ie, "hipGraphInstantiate", __FILE__, __LINE__);
        }
        if (rc != ADAS_OK) {  // nothing half-built stays behind
            if (g.graph) (void)hipGraphDestroy(g.graph);
            (void)hipGetLastError();
            return rc;
        }
        p->graphs.push_back(g);
        exec = g.exec;
    }
    ADAS_HIP_TRY(hipEventRecord(p->ev[0], p->st));
    ADAS_HIP_TRY(hipGraphLaunch(exec, p->st));
    ADAS_HIP_TRY(hipEventRecord(p->ev[5], p->st));
    p->timed = false;
    return ADAS_OK;
}

int adas_pipeline_step(adas_pipeline* p, const float* d_det, const float* d_lane) {
    ADAS_REQUIRE(p, ADAS_ERR_INVALID, "null pipeline");
    ADAS_REQUIRE(!p->d.detector || d_det, ADAS_ERR_INVALID, "detector input missing");
    ADAS_REQUIRE(!p->d.lane || d_lane, ADAS_ERR_INVALID, "lane input missing");
    if (!(p->d.use_graph & 1)) {
        p->timed = true;
        return record_step(p, d_det, d_lane, true);
    }
    return replay_step(p, adas_pipeline::GraphKey{d_det, d_lane, 0, 0, 0.0}, d_det, d_lane, nullptr);
}

int adas_pipeline_step_frames(adas_pipeline* p, const uint8_t* d_frames_bgr, int src_h, int src_w, double lane_crop_ratio) {
    ADAS_REQUIRE(p && d_frames_bgr && src_h > 0 && src_w > 0, ADAS_ERR_INVALID, "adas_pipeline_step_frames: bad argument");
    ADAS_REQUIRE(!p->d.lane || (lane_crop_ratio > 0.0 && lane_crop_ratio <= 1.0), ADAS_ERR_INVALID, "lane crop ratio must be in (0, 1]");
    const size_t S = (size_t)p->d.n_streams * (p->d.micro_batch > 1 ? p->d.micro_batch : 1);
    if (!p->det_in && !p->lane_in) {
        const char* env = getenv("ADAS_NO_PACKED_SEAM");
        p->packed = !(env && env[0] == '1') && (!p->d.detector || adas_engine_accepts_packed_input(p->d.detector)) &&
                    (!p->d.lane || adas_engine_accepts_packed_input(p->d.lane));
    }
    if (p->d.detector && !p->det_in) {
        int64_t is[4];
        adas_engine_input_shape(p->d.detector, is);
        ADAS_HIP_TRY(hipMalloc((void**)&p->det_in, S * (size_t)(is[1] * is[2] * is[3]) * 4));
    }
    if (p->d.lane && !p->lane_in) {
        int64_t is[4];
        adas_engine_input_shape(p->d.lane, is);
        ADAS_HIP_TRY(hipMalloc((void**)&p->lane_in, S * (size_t)(is[1] * is[2] * is[3]) * 4));
    }
    const adas_pipeline::FrameSrc fs{d_frames_bgr, src_h, src_w, lane_crop_ratio};
    if (!(p->d.use_graph & 1)) {
        p->fsrc = fs;
        p->timed = true;
        int rc = record_step(p, p->det_in, p->lane_in, true);
        p->fsrc.frames = nullptr;
        return rc;
    }
    return replay_step(p, adas_pipeline::GraphKey{d_frames_bgr, nullptr, src_h, src_w, lane_crop_ratio}, p->det_in, p->lane_in, &fs);
}

int adas_pipeline_step_frames_host(adas_pipeline* p, const uint8_t* h_frames_bgr, int src_h, int src_w, double lane_crop_ratio) {
    ADAS_REQUIRE(p && h_frames_bgr && src_h > 0 && src_w > 0, ADAS_ERR_INVALID, "adas_pipeline_step_frames_host: bad argument");
    const size_t bytes = (size_t)p->d.n_streams * (p->d.micro_batch > 1 ? p->d.micro_batch : 1) * src_h * src_w * 3;
    if (!p->st_copy) {
        ADAS_HIP_TRY(hipStreamCreateWithFlags(&p->st_copy, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            ADAS_HIP_TRY(hipEventCreateWithFlags(&p->ev_copied[k], hipEventDisableTiming));
            ADAS_HIP_TRY(hipEventCreateWithFlags(&p->ev_consumed[k], hipEventDisableTiming));
        }
    }
    if (bytes > p->host_stage_bytes) {  // (re)size both staging buffers; captured steps that read the old ones are dropped
        ADAS_HIP_TRY(hipStreamSynchronize(p->st));
        ADAS_HIP_TRY(hipStreamSynchronize(p->st_copy));
        drop_graphs(p);
        for (int k = 0; k < 2; ++k) {
            if (p->host_stage[k]) (void)hipFree(p->host_stage[k]);
            p->host_stage[k] = nullptr;
            ADAS_HIP_TRY(hipMalloc((void**)&p->host_stage[k], bytes));
        }
        p->host_stage_bytes = bytes;
        p->host_steps = 0;
    }
    const int k = (int)(p->host_steps & 1);
    if (p->host_steps >= 2) ADAS_HIP_TRY(hipStreamWaitEvent(p->st_copy, p->ev_consumed[k], 0));  // the step two back is done with buffer k
    ADAS_HIP_TRY(hipMemcpyAsync(p->host_stage[k], h_frames_bgr, bytes, hipMemcpyHostToDevice, p->st_copy));
    ADAS_HIP_TRY(hipEventRecord(p->ev_copied[k], p->st_copy));
    ADAS_HIP_TRY(hipStreamWaitEvent(p->st, p->ev_copied[k], 0));
    int rc = adas_pipeline_step_frames(p, p->host_stage[k], src_h, src_w, lane_crop_ratio);
    if (rc) return rc;
    ADAS_HIP_TRY(hipEventRecord(p->ev_consumed[k], p->st));
    ++p->host_steps;
    return ADAS_OK;
}

int adas_pipeline_wait_upload(adas_pipeline* p) {
    ADAS_REQUIRE(p, ADAS_ERR_INVALID, "null pipeline");
    if (p->st_copy) ADAS_HIP_TRY(hipStreamSynchronize(p->st_copy));
    return ADAS_OK;
}

int adas_pipeline_sync(adas_pipeline* p) {
    ADAS_REQUIRE(p, ADAS_ERR_INVALID, "null pipeline");
    ADAS_HIP_TRY(hipStreamSynchronize(p->st));
    // engines running the opt-in multi-layer launches (ADAS_ML=1) report a timed-out dependency wait here: the step's results are then
    // incomplete and the caller must not use them (adas_engine_ml_status returns at once for every other engine)
    const int frames = p->d.n_streams * (p->d.micro_batch > 1 ? p->d.micro_batch : 1);
    for (adas_engine* e : {p->d.detector, p->d.lane})
        if (e) {
            uint32_t w = 0;
            int rc = adas_engine_ml_status(e, frames, &w);
            if (rc != ADAS_OK) return rc;
        }
    return ADAS_OK;
}

int adas_pipeline_timings(adas_pipeline* p, float ms[6]) {
    ADAS_REQUIRE(p && ms, ADAS_ERR_INVALID, "null argument");
    ADAS_HIP_TRY(hipStreamSynchronize(p->st));
    for (int i = 0; i < 6; ++i) ms[i] = 0.f;
    ADAS_HIP_TRY(hipEventElapsedTime(&ms[5], p->ev[0], p->ev[5]));
    if (p->timed)
        for (int i = 0; i < 5; ++i) ADAS_HIP_TRY(hipEventElapsedTime(&ms[i], p->ev[i], p->ev[i + 1]));
    return ADAS_OK;
}
}
