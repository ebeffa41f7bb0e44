// pipeline.cpp -- the fused per-frame ADAS step (detector + NMS, lane + decode, tracker) for
// n_streams independent video streams, optionally replayed from a hipGraph.
// Mirrors what demo.py:261-281 drives per frame, minus UI; nothing returns to the host per step.
#include "engine.h"

struct adas_yolo_post;
struct adas_ufld_decode;
struct adas_bytetrack;

struct adas_pipeline {
    adas_pipeline_desc d;
    hipStream_t st = nullptr;
    hipStream_t st_lane = nullptr;  // graph mode: the lane branch is captured on its own stream so the two nets overlap
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    struct Cached {
        const float* det;
        const float* lane;
        hipGraph_t graph;
        hipGraphExec_t exec;
    };
    std::vector<Cached> graphs;  // one captured step per distinct (detector input, lane input) pair
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool timed = false;
};

using namespace adas;

// events == true: everything on one stream with section events (per-stage timing).
// events == false: the lane branch (net + decode) runs on st_lane, forked/joined with events, so the
// latency-bound detector layers and the MFMA-bound lane layers share the chip (independent work:
// demo.py:261-281 runs them back to back only because the reference is single-threaded Python).
static int record_step(adas_pipeline* p, const float* d_det, const float* d_lane, bool events) {
    const int S = p->d.n_streams;
    hipStream_t st = p->st;
    const bool fork = !events && p->d.detector && p->d.lane && !(p->d.use_graph & 2);
    hipStream_t sl = fork ? p->st_lane : st;
    int rc;
    if (events) ADAS_HIP_TRY(hipEventRecord(p->ev[0], st));
    if (fork) {
        ADAS_HIP_TRY(hipEventRecord(p->ev_fork, st));
        ADAS_HIP_TRY(hipStreamWaitEvent(sl, p->ev_fork, 0));
    }
    if (p->d.detector) {
        rc = adas_engine_infer_device(p->d.detector, d_det, S, st);
        if (rc) return rc;
        if (events) ADAS_HIP_TRY(hipEventRecord(p->ev[1], st));
        rc = adas_yolo_post_run(p->d.post, adas_engine_output_device(p->d.detector, 0), S, st);
        if (rc) return rc;
    } else if (events) ADAS_HIP_TRY(hipEventRecord(p->ev[1], st));
    if (events) ADAS_HIP_TRY(hipEventRecord(p->ev[2], st));
    if (p->d.lane) {
        rc = adas_engine_infer_device(p->d.lane, d_lane, S, sl);
        if (rc) return rc;
        if (events) ADAS_HIP_TRY(hipEventRecord(p->ev[3], st));
        const adas_engine* le = p->d.lane;
        size_t stride = (size_t)le->bufs[le->outs[0].buf].h * le->bufs[le->outs[0].buf].w * le->bufs[le->outs[0].buf].c;
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
        rc = adas_bytetrack_update_device(p->d.tracker, xy, sc, cl, cn, cap, 4, 2, S, st);
        if (rc) return rc;
    }
    if (fork) {
        ADAS_HIP_TRY(hipEventRecord(p->ev_join, sl));
        ADAS_HIP_TRY(hipStreamWaitEvent(st, p->ev_join, 0));
    }
    if (events) ADAS_HIP_TRY(hipEventRecord(p->ev[5], st));
    return ADAS_OK;
}

extern "C" {

int adas_pipeline_create(const adas_pipeline_desc* d, adas_pipeline** out) {
    ADAS_REQUIRE(d && out && d->n_streams > 0, ADAS_ERR_INVALID, "adas_pipeline_create: bad argument");
    ADAS_REQUIRE(d->detector || d->lane, ADAS_ERR_INVALID, "pipeline needs a detector and/or a lane engine");
    ADAS_REQUIRE(!d->detector || d->post, ADAS_ERR_INVALID, "detector engine needs a yolo_post handle");
    ADAS_REQUIRE(!d->lane || d->decode, ADAS_ERR_INVALID, "lane engine needs a ufld_decode handle");
    ADAS_REQUIRE(!d->lane || adas_engine_num_outputs(d->lane) == 4, ADAS_ERR_INVALID,
                 "Output dims is error, please check model. load %d channels not match 4.", d->lane ? adas_engine_num_outputs(d->lane) : 0);
    ADAS_REQUIRE(!d->detector || d->n_streams <= d->detector->max_batch, ADAS_ERR_INVALID, "n_streams exceeds detector max_batch");
    ADAS_REQUIRE(!d->lane || d->n_streams <= d->lane->max_batch, ADAS_ERR_INVALID, "n_streams exceeds lane max_batch");
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
    *out = p;
    return ADAS_OK;
}

int adas_pipeline_destroy(adas_pipeline* p) {
    if (!p) return ADAS_OK;
    for (auto& g : p->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    for (auto& e : p->ev)
        if (e) (void)hipEventDestroy(e);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    if (p->st_lane) (void)hipStreamDestroy(p->st_lane);
    if (p->st) (void)hipStreamDestroy(p->st);
    delete p;
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
    hipGraphExec_t exec = nullptr;
    for (auto& g : p->graphs)
        if (g.det == d_det && g.lane == d_lane) exec = g.exec;
    if (!exec) {
        ADAS_REQUIRE(p->graphs.size() < 64, ADAS_ERR_CAPACITY, "more than 64 distinct input buffers: reuse staging buffers with use_graph");
        adas_pipeline::Cached g{d_det, d_lane, nullptr, nullptr};
        ADAS_HIP_TRY(hipStreamBeginCapture(p->st, hipStreamCaptureModeThreadLocal));
        int rc = record_step(p, d_det, d_lane, false);
        hipError_t ce = hipStreamEndCapture(p->st, &g.graph);
        if (rc) return rc;
        if (ce != hipSuccess) return hip_fail(ce, "hipStreamEndCapture", __FILE__, __LINE__);
        ADAS_HIP_TRY(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
        p->graphs.push_back(g);
        exec = g.exec;
    }
    ADAS_HIP_TRY(hipEventRecord(p->ev[0], p->st));
    ADAS_HIP_TRY(hipGraphLaunch(exec, p->st));
    ADAS_HIP_TRY(hipEventRecord(p->ev[5], p->st));
    p->timed = false;
    return ADAS_OK;
}

int adas_pipeline_sync(adas_pipeline* p) {
    ADAS_REQUIRE(p, ADAS_ERR_INVALID, "null pipeline");
    ADAS_HIP_TRY(hipStreamSynchronize(p->st));
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
