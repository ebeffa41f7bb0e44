// engine.cpp -- HipEngine: loads an ADASHIP1 model container and runs it on one MI355X.
// Replaces EngineBase / OnnxEngine / TensorRTEngine (coreEngine.py:7-39,120-186): same surface
// (input shape, output shapes+names, inference on an NCHW tensor), plus a device-resident form.
// Memory plan: every graph buffer gets its own HBM allocation sized for max_batch frames (the nets
// are tiny against 288 GB); weights are packed once on the device into the compute type with K
// padded to 32 and Cout to 128 so the conv kernel needs no bounds checks on the weight side.
#include "engine.h"
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

using namespace adas;

static size_t elem_size(const adas_engine* e, const EngBuf& b) { return b.f32 ? 4 : (size_t)prec_esize(e->prec); }   // split precision: a (hi, lo) pair

static TView make_view(const adas_engine* e, int buf, int coff, int c) {
    const EngBuf& b = e->bufs[buf];
    TView v;
    v.p = b.d;
    v.cs = b.c;
    v.coff = coff;
    v.c = c;
    v.h = b.h;
    v.w = b.w;
    v.f32 = b.f32 ? 1 : 0;
    return v;
}

// does conv `ci` (with a projection shortcut link) take its shortcut into its own launch at this batch?
static bool ds_folded(const adas_engine* e, int ci, int batch) {
    const EngOp& c = e->ops[ci];
    if (c.ds_src < 0 || c.kernel != CONV_HALO || c.halo_bn > 0) return false;   // (a narrow-block packing runs on conv_halo only)
    const FileOp& o = c.f;
    const FileOp& d = e->ops[c.ds_src].f;
    // exactly launch_conv's order of choice: halo_rw and the stride-2 kernel come before conv_halo8 and know nothing of ds_w, and
    // conv_halo8 is asked with the conv's real residual view (the projection's output buffer)
    const TView in = make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), out = make_view(e, o.out_buf, o.out_coff, o.out_c);
    if (halo_rw_applicable(o.kh, o.kw, o.stride, o.pad, batch, in, out)) return false;
    if (halo_s2p_applicable(o.kh, o.kw, o.stride, o.pad, o.res_mode, batch, in, out)) return false;
    if (!halo8_applicable(o.kh, o.kw, o.stride, o.pad, batch, in, out, make_view(e, o.res_buf, o.res_coff, o.out_c), o.res_mode)) return false;
    return halo8_ds_applicable(o.kh, o.kw, o.stride, o.pad, batch, in, out, make_view(e, d.in_buf[0], d.in_coff[0], d.in_c[0]));
}

// is op `i` one of the three convs a fused C2f launch computes besides its cv1?
static bool in_c2f(const adas_engine* e, int i) {
    for (auto& q : e->ops)
        if (q.c2f[0] == i || q.c2f[1] == i || q.c2f[2] == i) return true;
    return false;
}

static bool is_c2f_tail(const adas_engine* e, int i) {   // the block's cv2: its output IS materialised
    for (auto& q : e->ops)
        if (q.c2f[2] == i) return true;
    return false;
}

// The ConvArgs engine_run_op launches op `i` (a plain OP_CONV: not a stem / pair / C2f launch) with at this batch.
static ConvArgs conv_args_of(const adas_engine* e, int i, int batch) {
    const EngOp& op = e->ops[i];
    const FileOp& o = op.f;
    unsigned char* wb = (unsigned char*)e->d_weights;
    ConvArgs a;
    a.in = make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]);
    a.out = make_view(e, o.out_buf, o.out_coff, o.out_c);
    if (o.res_mode != RES_NONE) a.res = make_view(e, o.res_buf, o.res_coff, o.out_c);
    else { a.res = a.out; a.res.p = nullptr; }
    a.wgt = wb + op.w_off;
    a.bias = (const float*)(wb + op.b_off);
    a.n = batch; a.kh = o.kh; a.kw = o.kw; a.stride = o.stride; a.pad = o.pad; a.act = o.act; a.res_mode = o.res_mode;
    a.k = op.k; a.kpad = op.kpad; a.m = batch * a.out.h * a.out.w; a.max_n = e->max_batch; a.prec = e->prec;
    if (op.has_x3h8) a.wgt_h8x3 = wb + op.x3h8_w_off;
    a.halo_bn = op.halo_bn;
    if (op.ds_src >= 0 && ds_folded(e, i, batch)) {
        const EngOp& dsop = e->ops[op.ds_src];
        a.ds_in = make_view(e, dsop.f.in_buf[0], dsop.f.in_coff[0], dsop.f.in_c[0]);
        a.ds_w = wb + dsop.ds_w_off;
        a.ds_bias = (const float*)(wb + dsop.b_off);
    }
    if (op.up_src >= 0) {
        const FileOp& u = e->ops[op.up_src].f;
        a.up = make_view(e, u.in_buf[0], u.in_coff[0], u.in_c[0]);
        a.up_c = (int)u.out_c;
    }
    return a;
}

// ---- multi-layer launches (conv_ml.hip): opt-in, ADAS_ML=1 when the engine is created.
static bool ml_enabled(const adas_engine* e) { return e->ml_on; }   // decided when the engine was created (ADAS_ML=1, 16-bit precisions)

// Is op `i` a conv that launches on its own at this batch AND has a tile body in the multi-layer kernel?
static bool ml_candidate(const adas_engine* e, int i, int batch, ConvArgs* out) {
    const EngOp& op = e->ops[i];
    const FileOp& o = op.f;
    if (o.type != OP_CONV || op.skip || (op.kernel != CONV_HALO && op.kernel != CONV_PW)) return false;
    if (op.pair_b >= 0 || op.c2f[0] >= 0 || op.fuse_pool >= 0 || op.fuse_conv2 >= 0) return false;
    if (op.ds_user >= 0 && ds_folded(e, op.ds_user, batch)) return false;   // launches nothing at this batch
    if (op.ds_src >= 0 && ds_folded(e, i, batch)) return false;              // carries its projection: conv_halo8 only
    auto aliased = [&](int b) { return b >= 0 && b < (int)e->buf_aliased.size() && e->buf_aliased[b]; };
    if (aliased(o.in_buf[0]) || aliased(o.out_buf) || (o.res_mode != RES_NONE && aliased(o.res_buf))) return false;
    if (op.up_src >= 0 && aliased(e->ops[op.up_src].f.in_buf[0])) return false;
    {   // experiments: ADAS_ML_ONLY=halo | pw keeps the other kind of layer out of the launches
        const char* only = getenv("ADAS_ML_ONLY");
        if (only && ((only[0] == 'h' && op.kernel != CONV_HALO) || (only[0] == 'p' && op.kernel != CONV_PW))) return false;
    }
    const ConvArgs a = conv_args_of(e, i, batch);
    if (!ml_layer_supported(a, op.kernel)) return false;
    if (out) *out = a;
    return true;
}

// Is op `i` a 3x3 conv that launches on conv_halo at this batch (a layer the grouped launch can carry)?
static bool group_candidate(const adas_engine* e, int i, int batch, ConvArgs* out) {
    const EngOp& op = e->ops[i];
    const FileOp& o = op.f;
    if (o.type != OP_CONV || op.skip || op.kernel != CONV_HALO) return false;
    if (op.pair_b >= 0 || op.c2f[0] >= 0 || op.fuse_pool >= 0 || op.fuse_conv2 >= 0 || op.up_src >= 0) return false;
    if (op.ds_user >= 0 && ds_folded(e, op.ds_user, batch)) return false;
    if (op.ds_src >= 0 && ds_folded(e, i, batch)) return false;
    auto aliased = [&](int b) { return b >= 0 && b < (int)e->buf_aliased.size() && e->buf_aliased[b]; };
    if (aliased(o.in_buf[0]) || aliased(o.out_buf) || (o.res_mode != RES_NONE && aliased(o.res_buf))) return false;
    const ConvArgs a = conv_args_of(e, i, batch);
    if (!group_layer_supported(a, op.kernel)) return false;
    if (out) *out = a;
    return true;
}

static const std::vector<GroupRun>* group_runs(const adas_engine* e, int batch) {
    auto it = e->groups.find(batch);
    return it == e->groups.end() ? nullptr : &it->second;
}

static const std::vector<MlSeg>* ml_segments(const adas_engine* e, int batch) {
    auto it = e->ml.find(batch);
    return it == e->ml.end() ? nullptr : &it->second;
}

static int free_engine(adas_engine* e) {
    if (!e) return ADAS_OK;
    for (auto& kv : e->ml)
        for (auto& sg : kv.second) ml_plan_destroy(sg.plan);
    e->ml.clear();
    for (auto& kv : e->groups)
        for (auto& run : kv.second)
            for (auto& st : run.steps) ml_group_destroy(st.group);
    e->groups.clear();
    for (auto& b : e->bufs)
        if (b.d && b.alias_of < 0) (void)hipFree(b.d);
    if (e->d_weights) (void)hipFree(e->d_weights);
    if (e->d_input) (void)hipFree(e->d_input);
    for (auto& ev : e->events)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : e->step_events)
        if (ev) (void)hipEventDestroy(ev);
    delete e;
    return ADAS_OK;
}

extern "C" {

int adas_engine_create(const char* model_path, int precision, int max_batch, adas_engine** out) {
    ADAS_REQUIRE(model_path && out && max_batch > 0, ADAS_ERR_INVALID, "adas_engine_create: bad argument");
    ADAS_REQUIRE(precision == ADAS_PREC_BF16 || precision == ADAS_PREC_FP32 || precision == ADAS_PREC_FP16 || precision == ADAS_PREC_FP16X3,
                 ADAS_ERR_INVALID, "unknown precision %d", precision);
    FILE* f = fopen(model_path, "rb");
    if (!f) {  // coreEngine.py:12-13
        set_error("The model path [%s] can't not found! (%s)", model_path, strerror(errno));
        return ADAS_ERR_IO;
    }
    FileHeader hd;
    if (fread(&hd, sizeof(hd), 1, f) != 1 || memcmp(hd.magic, "ADASHIP1", 8) != 0 || hd.version != 1) {
        fclose(f);
        set_error("[%s] is not an ADASHIP1 model container (convert an ONNX export with vehicle-cv-adas_amd/onnx_import.py or "
                  "build one with models.py; TensorRT plans cannot be imported)", model_path);
        return ADAS_ERR_FORMAT;
    }
    ADAS_REQUIRE(adas_device_count() > 0, (fclose(f), ADAS_ERR_NO_DEVICE), "no HIP device visible; this library has no CPU fallback");
    adas_engine* e = new adas_engine();
    e->prec = precision;
    e->max_batch = max_batch;
    {   // multi-layer launches are opt-in (ADAS_ML=1): measured slower than the per-layer launches at 64 frames (profiles/r05/ml_*.txt)
        const char* v = getenv("ADAS_ML");
        e->ml_on = prec_is16(precision) && v && v[0] == '1';
        const char* g = getenv("ADAS_NO_GROUP");
        e->group_on = prec_is16(precision) && !e->ml_on && !(g && g[0] == '1');
    }
    e->hdr = hd;
    e->name = std::string(hd.name, strnlen(hd.name, sizeof(hd.name)));
    std::vector<FileBuf> fb(hd.n_bufs);
    std::vector<FileOp> fo(hd.n_ops);
    std::vector<FileOut> fout(hd.n_outputs);
    bool ok = fread(fb.data(), sizeof(FileBuf), hd.n_bufs, f) == hd.n_bufs && fread(fo.data(), sizeof(FileOp), hd.n_ops, f) == hd.n_ops &&
              fread(fout.data(), sizeof(FileOut), hd.n_outputs, f) == hd.n_outputs;
    if (!ok) {
        fclose(f);
        free_engine(e);
        set_error("[%s]: truncated model container", model_path);
        return ADAS_ERR_FORMAT;
    }
    // ---- every buffer index an op or output names must exist (a damaged container must not index past e->bufs)
    {
        auto bad = [&](int64_t b) { return b < 0 || b >= (int64_t)hd.n_bufs; };
        const char* what = nullptr;
        for (auto& o : fo) {
            if (o.n_in > 8) { what = "more than 8 inputs"; break; }
            for (uint32_t k = 0; k < o.n_in; ++k)
                if (bad(o.in_buf[k])) what = "input buffer";
            if (bad(o.out_buf)) what = "output buffer";
            if (o.res_mode != RES_NONE && bad(o.res_buf)) what = "residual buffer";
            if (what) break;
        }
        for (auto& q : fout)
            if (bad(q.buf)) what = "graph output buffer";
        if (what) {
            fclose(f);
            free_engine(e);
            set_error("[%s]: %s index out of range (container has %u buffers)", model_path, what, hd.n_bufs);
            return ADAS_ERR_FORMAT;
        }
    }
    // ---- buffers
    for (auto& b : fb) {
        EngBuf eb;
        eb.h = b.h; eb.w = b.w; eb.c = b.c; eb.f32 = (b.flags & 1) != 0; eb.d = nullptr;
        eb.alias_of = (b.flags & 2) ? (int)(b.flags >> 8) : -1;
        e->bufs.push_back(eb);
    }
    for (size_t bi = 0; bi < e->bufs.size(); ++bi) {  // an alias re-declares the shape of an EARLIER buffer's memory (torch .view)
        EngBuf& b = e->bufs[bi];
        if (b.alias_of < 0) continue;
        const bool ok = b.alias_of < (int)bi && e->bufs[b.alias_of].alias_of < 0 &&
                        (size_t)b.h * b.w * b.c == (size_t)e->bufs[b.alias_of].h * e->bufs[b.alias_of].w * e->bufs[b.alias_of].c &&
                        b.f32 == e->bufs[b.alias_of].f32;
        if (!ok) {
            fclose(f);
            free_engine(e);
            set_error("[%s]: buffer %zu is not a valid alias", model_path, bi);
            return ADAS_ERR_FORMAT;
        }
    }
    if (precision == PREC_X3)   // the G8 layout groups 8 channels: every 16-bit tensor of the graph must be a whole number of groups
        for (size_t bi = 0; bi < e->bufs.size(); ++bi)
            if (!e->bufs[bi].f32 && (e->bufs[bi].c & 7)) {
                fclose(f);
                free_engine(e);
                set_error("[%s]: buffer %zu has %d channels: the split precision (fp16x3) needs multiples of 8", model_path, bi, e->bufs[bi].c);
                return ADAS_ERR_FORMAT;
            }
    for (auto& b : e->bufs) {
        if (b.alias_of >= 0) continue;
        size_t bytes = (size_t)max_batch * b.h * b.w * b.c * elem_size(e, b);
        if (hipMalloc(&b.d, bytes + 256) != hipSuccess) {
            fclose(f);
            free_engine(e);
            return hip_fail(hipGetLastError(), "hipMalloc(activation buffer)", __FILE__, __LINE__);
        }
        (void)hipMemset(b.d, 0, bytes + 256);
        e->act_bytes += bytes;
    }
    for (auto& b : e->bufs)
        if (b.alias_of >= 0) b.d = e->bufs[b.alias_of].d;
    e->buf_aliased.assign(e->bufs.size(), 0);
    for (size_t bi = 0; bi < e->bufs.size(); ++bi)
        if (e->bufs[bi].alias_of >= 0) e->buf_aliased[bi] = e->buf_aliased[e->bufs[bi].alias_of] = 1;
    // ---- the operators without a generic fallback must be shapes their kernel takes (a damaged or foreign container fails here, not at launch)
    for (auto& o : fo) {
        auto view = [&](int buf, int coff, int c) { return make_view(e, buf, coff, c); };
        bool ok = true;
        if (o.type == OP_DWCONV)
            ok = o.n_in == 1 && o.kh == o.kw &&
                 dwconv_supported((int)o.kh, (int)o.stride, (int)o.pad, (int)o.res_mode, view(o.in_buf[0], o.in_coff[0], o.in_c[0]), view(o.out_buf, o.out_coff, o.out_c)) &&
                 o.w_elems == (uint64_t)o.kh * o.kw * o.out_c && o.b_elems == (uint64_t)o.out_c;
        else if (o.type == OP_ATTENTION)
            ok = o.n_in == 1 && attention_supported((int)o.params[0], (int)o.params[1], (int)o.params[2], view(o.in_buf[0], o.in_coff[0], o.in_c[0]),
                                                    view(o.out_buf, o.out_coff, o.out_c));
        else if (o.type == OP_DEPTH2SPACE)
            ok = o.n_in == 1 && depth2space_supported(view(o.in_buf[0], o.in_coff[0], o.in_c[0]), view(o.out_buf, o.out_coff, o.out_c));
        else if (o.type == OP_DETECT_V6)
            ok = o.n_in == 6;
        else if (o.type == OP_SE_GATE)
            ok = o.n_in == 1 && se_gate_supported(view(o.in_buf[0], o.in_coff[0], o.in_c[0]), view(o.out_buf, o.out_coff, o.out_c), (int)o.params[0], o.w_elems, o.b_elems);
        else if (o.type == OP_SCALE)
            ok = o.n_in == 2 && scale_supported(view(o.in_buf[0], o.in_coff[0], o.in_c[0]), view(o.in_buf[1], o.in_coff[1], o.in_c[1]), view(o.out_buf, o.out_coff, o.out_c));
        else if (o.type == OP_SHUFFLE)
            ok = o.n_in == 1 && shuffle_supported(view(o.in_buf[0], o.in_coff[0], o.in_c[0]), view(o.out_buf, o.out_coff, o.out_c), (int)o.params[0]);
        else if (o.type == OP_WSUM) {
            TView ins[3];
            for (uint32_t k = 0; k < o.n_in && k < 3; ++k) ins[k] = view(o.in_buf[k], o.in_coff[k], o.in_c[k]);
            ok = o.n_in <= 3 && wsum_supported((int)o.n_in, ins, view(o.out_buf, o.out_coff, o.out_c)) && o.act <= ACT_RELU6;
        }
        if ((o.type == OP_CONV || o.type == OP_DWCONV) && o.act > ACT_LEAKY) {   // hard-swish / hard-sigmoid: element-wise layers only (kernels.h)
            fclose(f);
            free_engine(e);
            set_error("[%s]: layer %s: activation %u is not a convolution epilogue (lower it as a one-input weighted-sum layer)", model_path,
                      std::string(o.name, strnlen(o.name, sizeof(o.name))).c_str(), o.act);
            return ADAS_ERR_FORMAT;
        }
        if (!ok) {
            fclose(f);
            free_engine(e);
            set_error("[%s]: layer %s: unsupported %s shape", model_path, std::string(o.name, strnlen(o.name, sizeof(o.name))).c_str(),
                      o.type == OP_DWCONV ? "depth-wise convolution" : o.type == OP_ATTENTION ? "attention" : o.type == OP_DEPTH2SPACE ? "depth-to-space"
                      : o.type == OP_SE_GATE ? "squeeze-and-excitation" : o.type == OP_SCALE ? "channel scale" : o.type == OP_WSUM ? "weighted sum" : o.type == OP_SHUFFLE ? "channel shuffle" : "Detect");
            return ADAS_ERR_FORMAT;
        }
    }
    // ---- weights: stream the fp32 blob through a staging buffer, pack on the device
    size_t packed_total = 0;
    const size_t esz = (size_t)prec_esize(precision);
    for (auto& o : fo) {
        EngOp op;
        op.f = o;
        op.name = std::string(o.name, strnlen(o.name, sizeof(o.name)));
        op.w_off = op.b_off = 0;
        e->ops.push_back(op);
    }
    // The fusion passes below count the readers of a tensor by buffer index.  A buffer that is re-viewed through an alias
    // (Graph.alias: the same bytes under another shape) has readers those counts would miss, so such buffers stay out of every fusion.
    auto aliased = [&](int64_t buf) {
        if (buf < 0 || buf >= (int64_t)e->bufs.size()) return false;
        if (e->bufs[buf].alias_of >= 0) return true;
        for (auto& b : e->bufs)
            if (b.alias_of == (int)buf) return true;
        return false;
    };
    // ---- first-layer fusion (conv_stem.hip): input conversion + stride-2 conv (+ the ResNet stem's max-pool) in one launch
    {
        const char* env = getenv("ADAS_NO_STEM");
        const bool enabled = !(env && env[0] == '1');
        auto reads_buf = [&](const FileOp& q, int buf) {
            for (uint32_t k = 0; k < q.n_in && k < 8; ++k)
                if (q.in_buf[k] == buf) return true;
            return q.res_mode != RES_NONE && q.res_buf == buf;
        };
        auto is_output = [&](int buf) {
            for (auto& q : fout)
                if ((int)q.buf == buf) return true;
            return false;
        };
        if (enabled && precision == PREC_X3 && e->ops.size() >= 2 && fo[0].type == OP_INPUT && fo[1].type == OP_CONV && fo[1].in_buf[0] == fo[0].out_buf &&
            !aliased(fo[0].out_buf) && !aliased(fo[1].out_buf)) {
            // split precision: input conversion + first conv in one launch (conv_stem_x3.hip); the max-pool / second conv stay separate
            bool only = true;
            for (size_t i = 2; i < fo.size(); ++i) only = only && !reads_buf(fo[i], fo[0].out_buf);
            if (only && stem_x3_applicable(hd.in_c, fo[1].kh, fo[1].kw, fo[1].stride, fo[1].pad, fo[1].act, fo[1].res_mode,
                                           make_view(e, fo[1].out_buf, fo[1].out_coff, fo[1].out_c))) {
                e->ops[0].skip = true;
                e->ops[1].kernel = CONV_STEM;
                // the ResNet stem's max-pool joins the launch when nothing else reads the conv output
                bool pool = e->ops.size() >= 3 && fo[2].type == OP_MAXPOOL && fo[2].kh == 3 && fo[2].stride == 2 && fo[2].pad == 1 &&
                            fo[2].in_buf[0] == fo[1].out_buf && fo[2].in_coff[0] == fo[1].out_coff && fo[2].in_c[0] == fo[1].out_c &&
                            !is_output(fo[1].out_buf) && !aliased(fo[2].out_buf);
                for (size_t i = 3; i < fo.size() && pool; ++i) pool = !reads_buf(fo[i], fo[1].out_buf);
                const char* envp = getenv("ADAS_NO_STEM_POOL_X3");
                if (pool && !(envp && envp[0] == '1') &&
                    stem_pool_x3_applicable(hd.in_c, fo[1].kh, fo[1].kw, fo[1].stride, fo[1].pad, fo[1].act, fo[1].res_mode,
                                            make_view(e, fo[1].out_buf, fo[1].out_coff, fo[1].out_c), make_view(e, fo[2].out_buf, fo[2].out_coff, fo[2].out_c))) {
                    e->ops[1].fuse_pool = 2;
                    e->ops[2].skip = true;
                }
                // YOLO stems: the 3x3 s2 conv on the stem's 16 channels joins the launch when nothing else reads the stem output (conv_stem2_x3_kernel)
                if (e->ops[1].fuse_pool < 0 && e->ops.size() >= 3 && fo[2].type == OP_CONV && fo[2].n_in == 1 && fo[2].in_buf[0] == fo[1].out_buf &&
                    fo[2].in_coff[0] == fo[1].out_coff && fo[2].in_c[0] == fo[1].out_c && !is_output(fo[1].out_buf) && !aliased(fo[2].out_buf)) {
                    bool sole = true;
                    for (size_t i = 3; i < fo.size(); ++i) sole = sole && !reads_buf(fo[i], fo[1].out_buf);
                    if (sole && stem2_x3_applicable(hd.in_c, fo[1].kh, fo[1].pad, fo[1].act, make_view(e, fo[1].out_buf, fo[1].out_coff, fo[1].out_c), fo[2].kh,
                                                    fo[2].kw, fo[2].stride, fo[2].pad, fo[2].act, fo[2].res_mode, make_view(e, fo[2].out_buf, fo[2].out_coff, fo[2].out_c))) {
                        e->ops[1].fuse_conv2 = 2;
                        e->ops[2].skip = true;
                        e->ops[2].kernel = CONV_STEM2;
                    }
                }
            }
        } else if (enabled && e->ops.size() >= 2 && fo[0].type == OP_INPUT && fo[1].type == OP_CONV && fo[1].in_buf[0] == fo[0].out_buf &&
            !aliased(fo[0].out_buf) && !aliased(fo[1].out_buf)) {
            bool only = true;
            for (size_t i = 2; i < fo.size(); ++i) only = only && !reads_buf(fo[i], fo[0].out_buf);
            bool pool = e->ops.size() >= 3 && fo[2].type == OP_MAXPOOL && fo[2].kh == 3 && fo[2].stride == 2 && fo[2].pad == 1 &&
                        fo[2].in_buf[0] == fo[1].out_buf && fo[2].in_coff[0] == fo[1].out_coff && fo[2].in_c[0] == fo[1].out_c &&
                        !is_output(fo[1].out_buf);
            for (size_t i = 3; i < fo.size() && pool; ++i) pool = !reads_buf(fo[i], fo[1].out_buf);
            TView cv = make_view(e, fo[1].out_buf, fo[1].out_coff, fo[1].out_c);
            TView pv = pool ? make_view(e, fo[2].out_buf, fo[2].out_coff, fo[2].out_c) : cv;
            if (pool && !stem_applicable(precision, hd.in_c, fo[1].kh, fo[1].kw, fo[1].stride, fo[1].pad, fo[1].act, fo[1].res_mode, cv, true, pv)) pool = false;
            if (only && stem_applicable(precision, hd.in_c, fo[1].kh, fo[1].kw, fo[1].stride, fo[1].pad, fo[1].act, fo[1].res_mode, cv, pool, pool ? pv : cv)) {
                e->ops[0].skip = true;
                e->ops[1].kernel = CONV_STEM;
                if (pool) {
                    e->ops[1].fuse_pool = 2;
                    e->ops[2].skip = true;
                }
                // YOLO stems: the 3x3 s2 conv on the stem's 16 channels joins the launch when nothing else reads the stem output
                const char* env2 = getenv("ADAS_NO_STEM2");
                if (!pool && !(env2 && env2[0] == '1') && e->ops.size() >= 3 && fo[2].type == OP_CONV && fo[2].in_buf[0] == fo[1].out_buf &&
                    fo[2].in_coff[0] == fo[1].out_coff && fo[2].in_c[0] == fo[1].out_c && !is_output(fo[1].out_buf)) {
                    bool sole = true;
                    for (size_t i = 3; i < fo.size(); ++i) sole = sole && !reads_buf(fo[i], fo[1].out_buf);
                    TView o2 = make_view(e, fo[2].out_buf, fo[2].out_coff, fo[2].out_c);
                    if (sole && stem2_applicable(precision, fo[1].kh, fo[1].pad, fo[1].act, cv, fo[2].kh, fo[2].kw, fo[2].stride, fo[2].pad, fo[2].act,
                                                 fo[2].res_mode, o2)) {
                        e->ops[1].fuse_conv2 = 2;
                        e->ops[2].skip = true;
                        e->ops[2].kernel = CONV_STEM2;
                    }
                }
            }
        }
    }
    // ---- projection shortcut folded into the conv that adds it (ResNet layerN.0: conv2 + downsample): decided per launch, because
    // the kernel that can do it (conv_halo8.hip) is chosen by batch
    for (size_t ci = 0; ci < fo.size(); ++ci) {
        const FileOp& c = fo[ci];
        if (c.type != OP_CONV || c.kh != 3 || c.kw != 3 || c.stride != 1 || c.pad != 1 || c.res_mode != RES_BEFORE_ACT || !prec_is16(precision)) continue;
        int di = -1;
        for (int j = (int)ci - 1; j >= 0 && di < 0; --j)
            if (fo[j].type == OP_CONV && fo[j].out_buf == c.res_buf && fo[j].out_coff == c.res_coff && fo[j].out_c == c.out_c) di = j;
        if (di < 0) continue;
        const FileOp& d = fo[di];
        if (d.kh != 1 || d.kw != 1 || d.stride != 2 || d.pad != 0 || d.act != ACT_NONE || d.res_mode != RES_NONE || d.n_in != 1 || (d.in_c[0] & 31) ||
            (d.out_c & 63) || e->ops[di].skip)
            continue;
        int nread = 0;
        for (size_t j = 0; j < fo.size(); ++j) {
            bool r = fo[j].res_mode != RES_NONE && fo[j].res_buf == d.out_buf;
            for (uint32_t t = 0; t < fo[j].n_in && t < 8; ++t) r = r || fo[j].in_buf[t] == d.out_buf;
            nread += r ? 1 : 0;
        }
        bool is_out = false;
        for (auto& q : fout) is_out = is_out || q.buf == d.out_buf;
        bool clean = true;   // x is not rewritten between the projection and the conv
        for (int j = di + 1; j < (int)ci && clean; ++j) clean = fo[j].out_buf != d.in_buf[0];
        if (nread != 1 || is_out || !clean || aliased(d.out_buf)) continue;
        e->ops[ci].ds_src = di;
        e->ops[di].ds_user = (int)ci;
    }
    // ---- nearest 2x upsample folded into its consumer: the upsample writes the leading channels of a concat buffer that exactly one
    // 1x1 conv reads (YOLO necks: Upsample -> Concat -> C2f.cv1); that conv then fetches those channels from the half-resolution
    // tensor itself and the upsample launch (and its 4x larger copy of the tensor) disappears
    {
        const char* env = getenv("ADAS_NO_UPSAMPLE_FOLD");
        const bool enabled = (prec_is16(precision) || precision == PREC_X3) && !(env && env[0] == '1');
        for (size_t ui = 0; enabled && ui < fo.size(); ++ui) {
            const FileOp& u = fo[ui];
            if (u.type != OP_UPSAMPLE2 || e->ops[ui].skip || u.out_coff != 0 || (u.out_c & 31)) continue;
            int reader = -1, nread = 0;
            for (size_t j = 0; j < fo.size(); ++j) {
                if (j == ui) continue;
                // readers of the upsampled channel range (another slice of the same concat buffer may have its own readers)
                auto overlaps = [&](uint32_t buf, uint32_t coff, uint32_t c) { return buf == u.out_buf && coff < u.out_coff + u.out_c && coff + c > u.out_coff; };
                bool r = fo[j].res_mode != RES_NONE && overlaps(fo[j].res_buf, fo[j].res_coff, fo[j].out_c);
                for (uint32_t t = 0; t < fo[j].n_in && t < 8; ++t) r = r || overlaps(fo[j].in_buf[t], fo[j].in_coff[t], fo[j].in_c[t]);
                if (r) { ++nread; reader = (int)j; }
            }
            bool is_out = false;
            for (auto& q : fout) is_out = is_out || q.buf == u.out_buf;
            if (nread != 1 || is_out || reader <= (int)ui || aliased(u.out_buf)) continue;
            const FileOp& c = fo[reader];
            if (c.type != OP_CONV || c.kh != 1 || c.kw != 1 || c.stride != 1 || c.pad != 0 || c.res_mode != RES_NONE || c.n_in != 1 || c.in_coff[0] != 0 ||
                c.in_c[0] <= u.out_c || e->ops[reader].skip)
                continue;
            TView cin = make_view(e, c.in_buf[0], c.in_coff[0], c.in_c[0]), cout = make_view(e, c.out_buf, c.out_coff, c.out_c);
            // only conv_pw reads ConvArgs::up: fold when the reader is PLANNED onto it (ADAS_NO_PW=1 plans it elsewhere)
            if (plan_conv(precision, 1, 1, 1, 0, max_batch, RES_NONE, cin, cout).kernel != CONV_PW) continue;
            bool clean = true;   // the low-resolution source is not rewritten between the upsample and the conv
            for (int j = (int)ui + 1; j < reader && clean; ++j) clean = fo[j].out_buf != u.in_buf[0];
            if (!clean) continue;
            e->ops[reader].up_src = (int)ui;
            e->ops[ui].skip = true;
        }
    }
    // ---- SPPF: three chained 5x5 s1 p2 max-pools (each reading the previous one's output) run as one launch
    {
        const char* env = getenv("ADAS_NO_POOL_FUSE");
        const bool enabled = !(env && env[0] == '1');
        for (size_t i = 0; enabled && i + 2 < fo.size(); ++i) {
            const FileOp &p0 = fo[i], &p1 = fo[i + 1], &p2 = fo[i + 2];
            auto is5 = [](const FileOp& q) { return q.type == OP_MAXPOOL && q.kh == 5 && q.stride == 1 && q.pad == 2 && q.n_in == 1; };
            auto feeds = [](const FileOp& a, const FileOp& b) { return b.in_buf[0] == a.out_buf && b.in_coff[0] == a.out_coff && b.in_c[0] == a.out_c; };
            if (!is5(p0) || !is5(p1) || !is5(p2) || !feeds(p0, p1) || !feeds(p1, p2) || e->ops[i].skip) continue;
            TView in = make_view(e, p0.in_buf[0], p0.in_coff[0], p0.in_c[0]);
            TView outs[3] = {make_view(e, p0.out_buf, p0.out_coff, p0.out_c), make_view(e, p1.out_buf, p1.out_coff, p1.out_c),
                             make_view(e, p2.out_buf, p2.out_coff, p2.out_c)};
            if (!sppf_pool3_applicable(precision, in, outs)) continue;
            e->ops[i].pool3[0] = (int)i + 1;
            e->ops[i].pool3[1] = (int)i + 2;
            e->ops[i + 1].skip = e->ops[i + 2].skip = true;
            i += 2;
        }
        // SPP (YOLOv7's SPPCSPC, YOLOv3/v4): 5x5, 9x9 and 13x13 stride-1 max-pools of ONE tensor.  Stride-1 max-pools with -inf padding
        // compose exactly (a 9x9 window clipped to the image = the 5x5 max of 5x5 maxima), so the three are the SPPF chain's three
        // outputs and take the same launch -- bit-identical, and the 81 / 169 sequential loads per output of the generic kernel go away
        for (size_t i = 0; enabled && i + 2 < fo.size(); ++i) {
            const FileOp &p0 = fo[i], &p1 = fo[i + 1], &p2 = fo[i + 2];
            auto isk = [](const FileOp& q, uint32_t k) { return q.type == OP_MAXPOOL && q.kh == k && q.stride == 1 && q.pad == k / 2 && q.n_in == 1; };
            auto same_in = [](const FileOp& a, const FileOp& b) { return a.in_buf[0] == b.in_buf[0] && a.in_coff[0] == b.in_coff[0] && a.in_c[0] == b.in_c[0]; };
            if (!isk(p0, 5) || !isk(p1, 9) || !isk(p2, 13) || !same_in(p0, p1) || !same_in(p0, p2) || e->ops[i].skip || e->ops[i + 1].skip || e->ops[i + 2].skip) continue;
            TView in = make_view(e, p0.in_buf[0], p0.in_coff[0], p0.in_c[0]);
            TView outs[3] = {make_view(e, p0.out_buf, p0.out_coff, p0.out_c), make_view(e, p1.out_buf, p1.out_coff, p1.out_c),
                             make_view(e, p2.out_buf, p2.out_coff, p2.out_c)};
            if (!sppf_pool3_applicable(precision, in, outs)) continue;
            e->ops[i].pool3[0] = (int)i + 1;
            e->ops[i].pool3[1] = (int)i + 2;
            e->ops[i + 1].skip = e->ops[i + 2].skip = true;
            i += 2;
        }
    }
    // ---- 3x3 -> 3x3 pair fusion (conv_pair.hip): conv A's output feeds only conv B (the Bottleneck of YOLOv8's C2f blocks)
    std::vector<int> pair_of(e->ops.size(), -1);   // B -> A
    for (size_t ai = 0; ai + 1 < e->ops.size(); ++ai) {
        const FileOp& qa = fo[ai];
        if (qa.type != OP_CONV || e->ops[ai].skip || e->ops[ai].kernel == CONV_STEM || pair_of[ai] >= 0) continue;
        int bi = -1, readers = 0;
        for (size_t j = 0; j < fo.size(); ++j) {
            if (j == ai) continue;
            bool reads = false;
            for (uint32_t t = 0; t < fo[j].n_in && t < 8; ++t) reads = reads || fo[j].in_buf[t] == qa.out_buf;
            reads = reads || (fo[j].res_mode != RES_NONE && fo[j].res_buf == qa.out_buf);
            if (reads) { ++readers; bi = (int)j; }
        }
        bool is_out = false;
        for (auto& q : fout) is_out = is_out || q.buf == qa.out_buf;
        if (readers != 1 || is_out || bi <= (int)ai || aliased(qa.out_buf)) continue;
        const FileOp& qb = fo[bi];
        if (qb.type != OP_CONV || e->ops[bi].skip || qb.n_in != 1 || qb.in_buf[0] != qa.out_buf || qb.in_coff[0] != qa.out_coff || qb.in_c[0] != qa.out_c) continue;
        bool clean = true;   // nothing between A and B writes A's input or B's output region's buffer in a way the fusion would reorder
        for (int j = (int)ai + 1; j < bi && clean; ++j) clean = fo[j].out_buf != qa.in_buf[0] && fo[j].out_buf != qb.out_buf;
        if (!clean) continue;
        TView x = make_view(e, qa.in_buf[0], qa.in_coff[0], qa.in_c[0]), t = make_view(e, qa.out_buf, qa.out_coff, qa.out_c);
        TView y = make_view(e, qb.out_buf, qb.out_coff, qb.out_c);
        TView r2 = qb.res_mode != RES_NONE ? make_view(e, qb.res_buf, qb.res_coff, qb.out_c) : y;
        // the pair writes y while other workgroups still read x halos: y must not overlap x (same memory, intersecting channel ranges)
        if (y.p == x.p && y.coff < x.coff + x.c && x.coff < y.coff + y.c) continue;
        if (!pair_applicable(precision, qa.kh, qa.kw, qa.stride, qa.pad, qa.act, qa.res_mode, x, t, qb.kh, qb.kw, qb.stride, qb.pad, qb.act, qb.res_mode, y, r2) &&
            !pair_x3_candidate(precision, qa.kh, qa.kw, qa.stride, qa.pad, qa.act, qa.res_mode, x, t, qb.kh, qb.kw, qb.stride, qb.pad, qb.act, qb.res_mode, y))
            continue;
        e->ops[ai].pair_b = bi;
        e->ops[bi].skip = true;
        pair_of[bi] = (int)ai;
    }
    // ---- whole-C2f fusion (conv_c2f.hip): cv1 1x1 -> [split] -> fused 3x3 pair with shortcut -> cv2 1x1 over the concat, when the concat
    // buffer has no other reader: one launch, the concat is never written (YOLOv8n / YOLOv10n model.2)
    // ---- v5-layout Detect fusion (aux_kernels.hip detect_v5_fused_kernel): the per-level 1x1 convs feed only the decode; decided before
    // the weight layout because the fused launch wants per-anchor MFMA fragments (CONV_DET5)
    std::vector<int> det5_of(e->ops.size(), -1);    // conv -> its OP_DETECT_V5
    for (size_t di = 0; di < fo.size(); ++di) {
        const FileOp& dq = fo[di];
        if (dq.type != OP_DETECT_V5 || dq.n_in != 3) continue;
        int src[3];
        bool ok = true;
        for (int k = 0; k < 3 && ok; ++k) {
            src[k] = -1;
            for (size_t j = 0; j < di; ++j)
                if (fo[j].type == OP_CONV && fo[j].out_buf == dq.in_buf[k] && fo[j].out_coff == dq.in_coff[k] && fo[j].out_c == dq.in_c[k]) src[k] = (int)j;
            ok = src[k] >= 0;
            if (!ok) break;
            const FileOp& q = fo[src[k]];
            ok = q.kh == 1 && q.kw == 1 && q.stride == 1 && q.pad == 0 && q.act == ACT_NONE && q.res_mode == RES_NONE && q.n_in == 1 && !e->ops[src[k]].skip &&
                 det5_applicable(precision, (int)dq.params[0], make_view(e, q.in_buf[0], q.in_coff[0], q.in_c[0]), make_view(e, q.out_buf, q.out_coff, q.out_c));
            for (size_t j = 0; j < fo.size() && ok; ++j) {  // nobody else reads the logits
                if (j == di) continue;
                for (uint32_t t = 0; t < fo[j].n_in && t < 8; ++t) ok = ok && fo[j].in_buf[t] != q.out_buf;
                ok = ok && !(fo[j].res_mode != RES_NONE && fo[j].res_buf == q.out_buf);
            }
            for (auto& out : fout) ok = ok && out.buf != q.out_buf;
            ok = ok && !aliased(q.out_buf);
        }
        if (!ok) continue;
        for (int k = 0; k < 3; ++k) {
            e->ops[di].det_src[k] = src[k];
            e->ops[src[k]].skip = true;
            det5_of[src[k]] = (int)di;
        }
    }
    std::vector<int> c2f_role(e->ops.size(), 0);   // 1: cv1 (launches the block), 2: cv2
    for (size_t ai = 0; ai < e->ops.size(); ++ai) {
        const int bi = e->ops[ai].pair_b;
        if (bi < 0) continue;
        const FileOp &qa = fo[ai], &qb = fo[bi];
        const int cat = qa.in_buf[0];
        if (qa.in_c[0] != 16 || qb.out_buf != cat || qb.out_coff != qa.in_coff[0] + 16 || qb.res_mode != RES_AFTER_ACT || qa.in_coff[0] < 16 || aliased(cat)) continue;
        if (qb.res_buf != cat || qb.res_coff != qa.in_coff[0]) continue;   // conv B's shortcut must be the y1 slice conv A reads (the fused kernel adds THAT)
        int c1 = -1, c2 = -1, readers = 0;
        for (size_t j = 0; j < fo.size(); ++j) {
            const FileOp& q = fo[j];
            bool reads = q.res_mode != RES_NONE && q.res_buf == cat;
            for (uint32_t t = 0; t < q.n_in && t < 8; ++t) reads = reads || q.in_buf[t] == cat;
            if (reads) ++readers;
            if (q.type != OP_CONV || q.kh != 1 || q.kw != 1 || q.stride != 1 || q.pad != 0 || q.act != ACT_SILU || q.res_mode != RES_NONE || q.n_in != 1 ||
                e->ops[j].skip)
                continue;
            if ((int)j < (int)ai && q.out_buf == cat && q.out_coff == qa.in_coff[0] - 16 && q.out_c == 32 && q.in_c[0] == 32 && e->ops[j].up_src < 0) c1 = (int)j;
            if ((int)j > bi && q.in_buf[0] == cat && q.in_coff[0] == qa.in_coff[0] - 16 && q.in_c[0] == 48 && q.out_c == 32) c2 = (int)j;
        }
        bool is_out = false;
        for (auto& q : fout) is_out = is_out || (int)q.buf == cat;
        if (c1 < 0 || c2 < 0 || readers != 3 || is_out) continue;   // readers: conv A, conv B's shortcut, cv2
        bool sole_writers = true;   // nothing else writes into the concat buffer
        for (size_t j = 0; j < fo.size(); ++j) sole_writers = sole_writers && !(fo[j].out_buf == cat && (int)j != c1 && (int)j != bi);
        if (!sole_writers) continue;
        const FileOp &q1 = fo[c1], &q2 = fo[c2];
        {   // the fused launch runs cv2 at cv1's position: nothing between cv1 and cv2 other than conv A / conv B may touch cv2's output buffer or
            // rewrite the block input, and cv2's output must not overlap the input whose halos other workgroups are still reading
            bool safe = true;
            for (int j = c1 + 1; j < c2 && safe; ++j) {
                if (j == (int)ai || j == bi) continue;
                const FileOp& q = fo[j];
                bool touches = q.out_buf == q2.out_buf || q.out_buf == q1.in_buf[0] || (q.res_mode != RES_NONE && q.res_buf == q2.out_buf);
                for (uint32_t t = 0; t < q.n_in && t < 8; ++t) touches = touches || q.in_buf[t] == q2.out_buf;
                safe = !touches;
            }
            if (q2.out_buf == q1.in_buf[0] && q2.out_coff < q1.in_coff[0] + q1.in_c[0] && q1.in_coff[0] < q2.out_coff + q2.out_c) safe = false;
            if (!safe) continue;
            const EngBuf& xb = e->bufs[q1.in_buf[0]];   // conv_c2f.hip addresses the input with 31-bit byte offsets: decided here, at max_batch
            if ((double)max_batch * xb.h * xb.w * xb.c * 2.0 >= 2147483648.0) continue;
        }
        {
            const TView vx = make_view(e, q1.in_buf[0], q1.in_coff[0], q1.in_c[0]), v01 = make_view(e, q1.out_buf, q1.out_coff, q1.out_c);
            const TView vy1 = make_view(e, qa.in_buf[0], qa.in_coff[0], qa.in_c[0]), vy2 = make_view(e, qb.out_buf, qb.out_coff, qb.out_c);
            const TView vcat = make_view(e, q2.in_buf[0], q2.in_coff[0], q2.in_c[0]), vout = make_view(e, q2.out_buf, q2.out_coff, q2.out_c);
            if (!c2f16_applicable(precision, vx, v01, vy1, vy2, vcat, vout) && !c2f16_x3_applicable(precision, vx, v01, vy1, vy2, vcat, vout)) continue;
        }
        e->ops[c1].c2f[0] = (int)ai; e->ops[c1].c2f[1] = bi; e->ops[c1].c2f[2] = c2;
        e->ops[ai].skip = true;   // (conv B is skipped already: the pair launch is replaced as a whole)
        e->ops[c2].skip = true;
        c2f_role[c1] = 1; c2f_role[c2] = 2;
    }
    if (precision == PREC_X3)   // split precision: a pair exists only inside a fused C2f block (conv_c2f_x3.hip) -- release the others
        for (size_t ai = 0; ai < e->ops.size(); ++ai) {
            const int bi = e->ops[ai].pair_b;
            if (bi < 0 || e->ops[ai].skip) continue;      // (skip: absorbed into a C2f launch above)
            e->ops[ai].pair_b = -1;
            e->ops[bi].skip = false;
            pair_of[bi] = -1;
        }
    for (auto& op : e->ops) {
        const FileOp& o = op.f;
        if (o.type == OP_CONV && op.kernel == CONV_STEM) {
            op.k = o.kh * o.kw * o.in_c[0];
            op.kpad = 32 * o.kh;
            op.cin_pad = 4;
            op.cout_pad = (o.out_c + 127) / 128 * 128;
            op.w_off = packed_total;
            packed_total += ((precision == PREC_X3 ? stem_x3_weight_bytes(o.kh, o.out_c) : stem_weight_bytes(o.kh, o.out_c)) + 255) & ~(size_t)255;
            op.b_off = packed_total;
            packed_total += ((size_t)op.cout_pad * 4 + 255) & ~(size_t)255;
        } else if (o.type == OP_CONV && op.kernel == CONV_STEM2) {
            op.k = o.kh * o.kw * o.in_c[0];
            op.kpad = 160;
            op.cin_pad = 16;
            op.cout_pad = (o.out_c + 127) / 128 * 128;
            op.w_off = packed_total;
            packed_total += ((precision == PREC_X3 ? stem2_x3_weight_bytes() : stem2_weight_bytes()) + 255) & ~(size_t)255;
            op.b_off = packed_total;
            packed_total += ((size_t)op.cout_pad * 4 + 255) & ~(size_t)255;
        } else if (o.type == OP_CONV) {
            int cin = o.in_c[0], cout = o.out_c;
            op.k = o.kh * o.kw * cin;
            ConvPlan pl = plan_conv(precision, o.kh, o.kw, o.stride, o.pad, max_batch, o.res_mode, make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]),
                                    make_view(e, o.out_buf, o.out_coff, o.out_c));
            if (pl.kernel == CONV_FC && o.res_mode != RES_NONE) {
                fclose(f);
                free_engine(e);
                set_error("[%s]: layer %s: a Linear layer cannot carry a residual", model_path, op.name.c_str());
                return ADAS_ERR_FORMAT;
            }
            op.kernel = pl.kernel;
            if (pl.kernel == CONV_HALO)   // few tiles at this engine's max_batch: narrower channel blocks (fixes the packing: decided here)
                op.halo_bn = plan_halo_bn(max_batch, (int)o.stride, make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), make_view(e, o.out_buf, o.out_coff, o.out_c));
            op.kpad = pl.kpad;
            op.cin_pad = pl.cin_pad;
            op.cout_pad = (cout + 127) / 128 * 128;
            const size_t self = (size_t)(&op - &e->ops[0]);
            if (op.pair_b >= 0 || pair_of[self] >= 0) op.kernel = CONV_PAIR;   // fragment packing (fits the plan's allocation: <= 18 KB)
            if (c2f_role[self]) op.kernel = CONV_C2F_PW;                        // 1x1 fragments: 2 / 4 KB, inside the plan's 8 / 16 KB
            if (det5_of[self] >= 0) {                                           // per-anchor fragments: 3 x 96 rows, more than the plan's 256
                op.kernel = CONV_DET5;
                op.w_off = packed_total;
                packed_total += (det5_weight_bytes(cout / 3, cin) + 255) & ~(size_t)255;
                op.b_off = packed_total;
                packed_total += ((size_t)op.cout_pad * 4 + 255) & ~(size_t)255;
                continue;
            }
            if (precision == PREC_X3 && pl.kernel != CONV_PW && pl.kernel != CONV_FC &&
                (halo8_x3_shape_ok(o.kh, o.kw, o.stride, o.pad, make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), make_view(e, o.out_buf, o.out_coff, o.out_c)) ||
                 halo_s2p_x3_shape_ok(o.kh, o.kw, o.stride, o.pad, o.res_mode, make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]),
                                      make_view(e, o.out_buf, o.out_coff, o.out_c)))) {
                op.has_x3h8 = true;   // the batch decides at launch which of the two packings runs
                op.x3h8_w_off = packed_total;
                packed_total += (halo8_x3_weight_bytes(cout, cin) + 255) & ~(size_t)255;
            }
            if (op.ds_user >= 0) {   // second copy of the projection weights, as per-step tiles
                op.ds_w_off = packed_total;
                packed_total += ((size_t)cout * cin * esz + 255) & ~(size_t)255;
            }
            op.w_off = packed_total;
            packed_total += ((size_t)op.cout_pad * op.kpad * esz + 255) & ~(size_t)255;
            op.b_off = packed_total;
            packed_total += ((size_t)op.cout_pad * 4 + 255) & ~(size_t)255;
        } else if (o.type == OP_LAYERNORM || o.type == OP_DWCONV || o.type == OP_SE_GATE) {
            op.w_off = packed_total;
            packed_total += ((size_t)o.w_elems * 4 + 255) & ~(size_t)255;
            op.b_off = packed_total;
            packed_total += ((size_t)o.b_elems * 4 + 255) & ~(size_t)255;
        } else if (o.type == OP_DETECT_V5) {
            op.w_off = packed_total;
            packed_total += 256;
        }
    }
    // ---- Detect fusion (aux_kernels.hip detect_v8_fused_kernel): the last 1x1 convs of both head branches feed only the decode
    {
        const char* env = getenv("ADAS_NO_DETECT_FUSE");
        const bool enabled = (prec_is16(precision) || precision == PREC_X3) && !(env && env[0] == '1');   // split precision: detect_v8_fused_x3_kernel
        for (size_t di = 0; enabled && di < e->ops.size(); ++di) {
            EngOp& dop = e->ops[di];
            if (dop.f.type != OP_DETECT_V8 || dop.f.n_in != 6) continue;
            int src[6];
            bool ok = true;
            for (int k = 0; k < 6 && ok; ++k) {
                src[k] = -1;
                for (size_t j = 0; j < di; ++j) {
                    const FileOp& q = e->ops[j].f;
                    if (q.type == OP_CONV && q.out_buf == dop.f.in_buf[k] && q.out_coff == dop.f.in_coff[k] && q.out_c == dop.f.in_c[k]) src[k] = (int)j;
                }
                ok = src[k] >= 0;
                if (!ok) break;
                const EngOp& c = e->ops[src[k]];
                const FileOp& q = c.f;
                ok = q.kh == 1 && q.kw == 1 && q.stride == 1 && q.act == ACT_NONE && q.res_mode == RES_NONE && c.kernel == CONV_PW && !c.skip &&
                     e->bufs[q.out_buf].f32 && !e->bufs[q.in_buf[0]].f32 && (q.in_c[0] & 7) == 0 && q.out_c == (k % 2 == 0 ? 64u : (uint32_t)dop.f.params[0]);
                for (size_t j = 0; j < e->ops.size() && ok; ++j) {  // nobody else reads the logits
                    if (j == di) continue;
                    const FileOp& r = e->ops[j].f;
                    for (uint32_t t = 0; t < r.n_in && t < 8; ++t) ok = ok && r.in_buf[t] != q.out_buf;
                    ok = ok && !(r.res_mode != RES_NONE && r.res_buf == q.out_buf);
                }
                for (auto& out : fout) ok = ok && out.buf != q.out_buf;
                ok = ok && !aliased(q.out_buf);
            }
            for (int l = 1; l < 3 && ok; ++l)  // one hidden width per branch
                ok = e->ops[src[2 * l]].f.in_c[0] == e->ops[src[0]].f.in_c[0] && e->ops[src[2 * l + 1]].f.in_c[0] == e->ops[src[1]].f.in_c[0];
            if (ok) {  // the fused launch keeps both weight matrices in LDS: leave very wide heads / class counts to the separate kernels
                const size_t ksb = (e->ops[src[0]].f.in_c[0] + 31) / 32, ksc = (e->ops[src[1]].f.in_c[0] + 31) / 32;
                const size_t ntc = ((size_t)dop.f.params[0] + 15) / 16;
                const size_t frag = precision == PREC_X3 ? 2048 : 1024;   // a 16x32 weight fragment: halves, or (hi, lo) half pairs
                if (ksc > 12 || (4 * ksb + ntc * ksc) * frag + (64 + ntc * 16) * 4 > 150 * 1024) ok = false;
                if (precision == PREC_X3)   // the fused kernel indexes conv_pw_x3's packing: [16-feature tile][kpad / 32]
                    for (int k = 0; k < 6 && ok; ++k) {
                        const EngOp& c = e->ops[src[k]];
                        ok = c.kpad == e->ops[src[k % 2]].kpad && (size_t)c.kpad >= (k % 2 ? ksc : ksb) * 32 && (c.kpad & 31) == 0 &&
                             (size_t)c.cout_pad >= (k % 2 ? ntc * 16 : 64);
                    }
                if (precision == PREC_X3 && ok && (size_t)e->ops[src[1]].kpad / 32 > 12) ok = false;
            }
            if (!ok) continue;
            for (int k = 0; k < 6; ++k) {
                dop.det_src[k] = src[k];
                e->ops[src[k]].skip = true;
            }
        }
    }
    e->weight_bytes = packed_total;
    if (hipMalloc(&e->d_weights, packed_total + 256) != hipSuccess) {
        fclose(f);
        free_engine(e);
        return hip_fail(hipGetLastError(), "hipMalloc(weights)", __FILE__, __LINE__);
    }
    (void)hipMemset(e->d_weights, 0, packed_total + 256);
    size_t max_w = 0;
    for (auto& o : fo) {
        max_w = o.w_elems > max_w ? (size_t)o.w_elems : max_w;
        max_w = o.b_elems > max_w ? (size_t)o.b_elems : max_w;   // layernorm / squeeze-and-excitation stage their second blob too
    }
    float* d_stage = nullptr;
    std::vector<float> h_stage(max_w ? max_w : 1);
    if (hipMalloc((void**)&d_stage, max_w * 4 + 256) != hipSuccess) {
        fclose(f);
        free_engine(e);
        return hip_fail(hipGetLastError(), "hipMalloc(weight staging)", __FILE__, __LINE__);
    }
    int rc = ADAS_OK;
    for (auto& op : e->ops) {
        const FileOp& o = op.f;
        unsigned char* base = (unsigned char*)e->d_weights;
        auto read_blob = [&](uint64_t off, uint64_t elems, float* dst) -> bool {
            if (fseek(f, (long)(hd.weights_off + off), SEEK_SET) != 0) return false;
            return fread(dst, 4, elems, f) == elems;
        };
        if (o.type == OP_CONV) {
            if (!read_blob(o.w_off, o.w_elems, h_stage.data()) || o.w_elems != (uint64_t)o.out_c * op.k) { rc = ADAS_ERR_FORMAT; break; }
            if (hipMemcpy(d_stage, h_stage.data(), o.w_elems * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = ADAS_ERR_HIP; break; }
            if (op.kernel == CONV_STEM || op.kernel == CONV_STEM2) {
                std::vector<uint16_t> frag((op.kernel == CONV_STEM2 ? (precision == PREC_X3 ? stem2_x3_weight_bytes() : stem2_weight_bytes())
                                            : precision == PREC_X3   ? stem_x3_weight_bytes(o.kh, o.out_c)
                                                                     : stem_weight_bytes(o.kh, o.out_c)) / 2);
                if (op.kernel == CONV_STEM2 && precision == PREC_X3) stem2_x3_pack_weights(h_stage.data(), frag.data());
                else if (op.kernel == CONV_STEM2) stem2_pack_weights(h_stage.data(), frag.data(), precision);
                else if (precision == PREC_X3) stem_x3_pack_weights(h_stage.data(), o.out_c, o.kh, o.kw, o.in_c[0], hd.in_c, frag.data());
                else stem_pack_weights(h_stage.data(), o.out_c, o.kh, o.kw, o.in_c[0], hd.in_c, frag.data(), precision);
                if (hipMemcpy(base + op.w_off, frag.data(), frag.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { rc = ADAS_ERR_HIP; break; }
                std::vector<float> b(op.cout_pad, 0.f);
                if (!read_blob(o.b_off, o.b_elems, b.data())) { rc = ADAS_ERR_FORMAT; break; }
                if (hipMemcpy(base + op.b_off, b.data(), (size_t)op.cout_pad * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = ADAS_ERR_HIP; break; }
                continue;
            }
            hipError_t pe = op.kernel == CONV_DET5 ? launch_pack_weights_det5(d_stage, base + op.w_off, o.out_c / 3, o.in_c[0], precision, 0)
                            : op.kernel == CONV_C2F_PW ? (precision == PREC_X3 ? launch_pack_weights_c2f_pw_x3(d_stage, base + op.w_off, o.out_c, o.in_c[0], 0)
                                                                                : launch_pack_weights_c2f_pw(d_stage, base + op.w_off, o.out_c, o.in_c[0], precision, 0))
                            : op.kernel == CONV_PAIR ? (precision == PREC_X3 ? launch_pack_weights_pair16_x3(d_stage, base + op.w_off, 0)
                                                                              : launch_pack_weights_pair(d_stage, base + op.w_off, o.out_c, precision, 0))
                            : (op.kernel == CONV_FC || op.kernel == CONV_PW)
                                ? launch_pack_weights_fc(d_stage, base + op.w_off, o.out_c, op.cout_pad, o.in_c[0], op.kpad, precision, 0)
                                : op.kernel == CONV_HALO
                                ? launch_pack_weights_halo(d_stage, base + op.w_off, o.out_c, op.cout_pad, o.in_c[0], op.cin_pad, precision, 0, op.halo_bn)
                                : launch_pack_weights(d_stage, base + op.w_off, o.out_c, op.cout_pad, o.kh * o.kw, o.in_c[0], op.cin_pad, op.kpad, precision, 0);
            if (pe == hipSuccess && op.has_x3h8) pe = launch_pack_weights_h8x3(d_stage, base + op.x3h8_w_off, o.out_c, o.in_c[0], 0);
            if (pe == hipSuccess && op.ds_user >= 0) pe = launch_pack_weights_ds(d_stage, base + op.ds_w_off, o.out_c, o.in_c[0], precision, 0);
            if (pe != hipSuccess) { rc = ADAS_ERR_HIP; break; }
            if (hipDeviceSynchronize() != hipSuccess) { rc = ADAS_ERR_HIP; break; }
            std::vector<float> b(op.cout_pad, 0.f);
            if (!read_blob(o.b_off, o.b_elems, b.data())) { rc = ADAS_ERR_FORMAT; break; }
            if (hipMemcpy(base + op.b_off, b.data(), (size_t)op.cout_pad * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = ADAS_ERR_HIP; break; }
        } else if (o.type == OP_LAYERNORM || o.type == OP_SE_GATE) {
            if (!read_blob(o.w_off, o.w_elems, h_stage.data())) { rc = ADAS_ERR_FORMAT; break; }
            if (hipMemcpy(base + op.w_off, h_stage.data(), o.w_elems * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = ADAS_ERR_HIP; break; }
            if (!read_blob(o.b_off, o.b_elems, h_stage.data())) { rc = ADAS_ERR_FORMAT; break; }
            if (hipMemcpy(base + op.b_off, h_stage.data(), o.b_elems * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = ADAS_ERR_HIP; break; }
        } else if (o.type == OP_DWCONV) {   // container [C][kh][kw] -> device [kh*kw][C] fp32 (every precision: 36..6272 floats per layer)
            const size_t C_ = o.out_c, T_ = (size_t)o.kh * o.kw;
            if (!read_blob(o.w_off, o.w_elems, h_stage.data())) { rc = ADAS_ERR_FORMAT; break; }
            std::vector<float> wt(C_ * T_);
            for (size_t c = 0; c < C_; ++c)
                for (size_t t = 0; t < T_; ++t) wt[t * C_ + c] = h_stage[c * T_ + t];
            if (hipMemcpy(base + op.w_off, wt.data(), wt.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = ADAS_ERR_HIP; break; }
            if (!read_blob(o.b_off, o.b_elems, h_stage.data())) { rc = ADAS_ERR_FORMAT; break; }
            if (hipMemcpy(base + op.b_off, h_stage.data(), o.b_elems * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = ADAS_ERR_HIP; break; }
        } else if (o.type == OP_DETECT_V5) {
            float anc[18];
            if (o.w_elems != 18 || !read_blob(o.w_off, 18, anc)) { rc = ADAS_ERR_FORMAT; break; }
            if (hipMemcpy(base + op.w_off, anc, sizeof(anc), hipMemcpyHostToDevice) != hipSuccess) { rc = ADAS_ERR_HIP; break; }
        }
    }
    (void)hipFree(d_stage);
    fclose(f);
    if (rc != ADAS_OK) {
        set_error("[%s]: failed while loading weights (%s)", model_path, rc == ADAS_ERR_HIP ? hipGetErrorString(hipGetLastError()) : "bad blob");
        free_engine(e);
        return rc;
    }
    // ---- outputs
    for (auto& o : fout) {
        EngOut eo;
        eo.buf = o.buf; eo.offset = o.offset; eo.ndim = o.ndim;
        for (int i = 0; i < 4; ++i) eo.dims[i] = o.dims[i];
        eo.elems = 1;
        for (int i = 1; i < (int)o.ndim; ++i) eo.elems *= o.dims[i];
        eo.name = std::string(o.name, strnlen(o.name, sizeof(o.name)));
        if (o.buf >= e->bufs.size() || !e->bufs[o.buf].f32) {
            set_error("[%s]: output %s is not an fp32 buffer", model_path, eo.name.c_str());
            free_engine(e);
            return ADAS_ERR_FORMAT;
        }
        e->outs.push_back(eo);
    }
    size_t in_bytes = (size_t)max_batch * hd.in_c * hd.in_h * hd.in_w * 4;
    if (hipMalloc((void**)&e->d_input, in_bytes) != hipSuccess) {
        free_engine(e);
        return hip_fail(hipGetLastError(), "hipMalloc(input staging)", __FILE__, __LINE__);
    }
    // the grouped / multi-layer launch tables of the engine's own batch size are built now, not on the first forward (device
    // allocations and synchronous copies do not belong on the hot path; other batch sizes are prepared on first use, engine_forward)
    if (engine_prepare(e, max_batch) != ADAS_OK) {
        free_engine(e);
        return ADAS_ERR_HIP;
    }
    *out = e;
    return ADAS_OK;
}

int adas_engine_destroy(adas_engine* e) { return free_engine(e); }

int adas_engine_input_shape(const adas_engine* e, int64_t dims[4]) {
    ADAS_REQUIRE(e && dims, ADAS_ERR_INVALID, "null argument");
    dims[0] = 1; dims[1] = e->hdr.in_c; dims[2] = e->hdr.in_h; dims[3] = e->hdr.in_w;
    return ADAS_OK;
}
int adas_engine_num_outputs(const adas_engine* e) { return e ? (int)e->outs.size() : 0; }
int adas_engine_output_shape(const adas_engine* e, int index, int64_t dims[4], int* ndim) {
    ADAS_REQUIRE(e && dims && ndim && index >= 0 && index < (int)e->outs.size(), ADAS_ERR_INVALID, "bad output index");
    for (int i = 0; i < 4; ++i) dims[i] = e->outs[index].dims[i];
    *ndim = e->outs[index].ndim;
    return ADAS_OK;
}
const char* adas_engine_output_name(const adas_engine* e, int index) {
    if (!e || index < 0 || index >= (int)e->outs.size()) return "";
    return e->outs[index].name.c_str();
}
int adas_engine_stats(const adas_engine* e, double* flops, double* wbytes, int* nl) {
    ADAS_REQUIRE(e, ADAS_ERR_INVALID, "null engine");
    if (flops) *flops = e->hdr.flops;
    if (wbytes) *wbytes = (double)e->weight_bytes;
    if (nl) *nl = (int)e->ops.size();
    return ADAS_OK;
}
int adas_engine_layer_kernel(const adas_engine* e, int layer, int batch, char* name, int cap) {
    ADAS_REQUIRE(e && name && cap > 0 && layer >= 0 && layer < (int)e->ops.size() && batch > 0, ADAS_ERR_INVALID, "bad layer index");
    const EngOp& op = e->ops[layer];
    const FileOp& o = op.f;
    static const char* kOther[] = {"input_nchw_kernel", "", "maxpool_kernel", "upsample2_kernel", "detect_v8_kernel", "detect_v5_kernel",
                                   "layernorm_kernel", "dwconv_kernel", "attention_kernel", "avgpool_kernel", "depth2space_kernel", "detect_v6_kernel",
                                   "se_gate_kernel", "scale_kernel", "wsum_kernel", "shuffle_kernel"};
    const MlSeg* in_seg = nullptr;
    if (const std::vector<MlSeg>* segs = ml_segments(e, batch))
        for (auto& sg : *segs)
            for (int m : sg.ops)
                if (m == layer) in_seg = &sg;
    const GroupRun* in_run = nullptr;
    if (const std::vector<GroupRun>* runs = group_runs(e, batch))
        for (auto& run : *runs)
            for (int m : run.ops)
                if (m == layer) in_run = &run;
    // a layer of a run belongs to one STEP of it: a grouped launch (its first member carries the label, the others ride in it) or a
    // single layer on its own kernel, which is labelled like any other layer below
    const GroupStep* in_step = nullptr;
    if (in_run)
        for (auto& st : in_run->steps)
            for (int m : st.members)
                if (m == layer) in_step = &st;
    if (in_step && in_step->group && in_step->members.front() == layer) {
        snprintf(name, cap, "conv_halo_group_kernel[%d layers]", (int)in_step->members.size());
    } else if (in_step && in_step->group) {
        snprintf(name, cap, "(in the grouped launch)");
    } else if (in_seg && in_seg->first == layer) {
        snprintf(name, cap, "conv_ml_kernel[%d layers]", in_seg->n_layers);
    } else if (in_seg) {
        snprintf(name, cap, "(in the multi-layer launch)");
    } else if (o.type == OP_CONV && op.ds_user >= 0 && ds_folded(e, op.ds_user, batch)) {
        snprintf(name, cap, "(fused into the conv it is the shortcut of)");
    } else if (op.skip && o.type == OP_UPSAMPLE2) {
        snprintf(name, cap, "(folded into the consumer's loads)");
    } else if (op.skip && o.type == OP_MAXPOOL && (o.kh == 5 || o.kh == 9 || o.kh == 13)) {
        snprintf(name, cap, "(fused into the SPPF pool launch)");
    } else if (o.type == OP_MAXPOOL && op.pool3[0] >= 0) {
        snprintf(name, cap, "sppf_pool3_kernel");
    } else if (o.type == OP_CONV && op.c2f[0] >= 0) {
        snprintf(name, cap, e->prec == PREC_X3 ? "conv_c2f16_x3_kernel" : "conv_c2f16_kernel");
    } else if (op.skip && o.type == OP_CONV && in_c2f(e, layer)) {
        snprintf(name, cap, "(fused into the C2f launch)");
    } else if (op.skip && op.kernel == CONV_PAIR) {
        snprintf(name, cap, "(fused into the pair launch)");
    } else if (o.type == OP_CONV && op.pair_b >= 0) {
        snprintf(name, cap, "conv_pair_kernel<%d>", (int)o.out_c);
    } else if (op.skip) {
        snprintf(name, cap, o.type == OP_CONV && o.kh == 1 && (op.kernel == CONV_PW || op.kernel == CONV_DET5) ? "(fused into the Detect launch)" : "(fused into the stem launch)");
    } else if (o.type == OP_CONV && op.kernel == CONV_STEM && op.fuse_conv2 >= 0) {
        snprintf(name, cap, e->prec == PREC_X3 ? "conv_stem2_x3_kernel<%d>+conv3x3s2" : "conv_stem_kernel<%d,1,SILU>+conv3x3s2", (int)o.kh);
    } else if (o.type == OP_CONV) {
        ConvArgs a;
        a.in = make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]);
        a.out = make_view(e, o.out_buf, o.out_coff, o.out_c);
        a.n = batch; a.kh = o.kh; a.kw = o.kw; a.stride = o.stride; a.pad = o.pad; a.act = o.act; a.res_mode = o.res_mode;
        if (o.res_mode != RES_NONE) a.res = make_view(e, o.res_buf, o.res_coff, o.out_c);
        else { a.res = a.out; a.res.p = nullptr; }
        a.k = op.k; a.kpad = op.kpad; a.m = batch * a.out.h * a.out.w; a.max_n = e->max_batch; a.prec = e->prec;
        if (op.has_x3h8) a.wgt_h8x3 = (const unsigned char*)e->d_weights + op.x3h8_w_off;
        a.halo_bn = op.halo_bn;
        snprintf(name, cap, "%s%s%s", conv_kernel_name(a, e->prec, op.kernel), op.fuse_pool >= 0 ? "+pool" : "",
                 (op.ds_src >= 0 && ds_folded(e, layer, batch)) ? "+shortcut" : "");
    } else if (o.type == OP_DETECT_V8 && op.det_src[0] >= 0) {
        snprintf(name, cap, e->prec == PREC_X3 ? "detect_v8_fused_x3_kernel" : "detect_v8_fused_kernel");
    } else if (o.type == OP_DETECT_V5 && op.det_src[0] >= 0) {
        snprintf(name, cap, "detect_v5_fused_kernel");
    } else {
        snprintf(name, cap, "%s", o.type < 16 ? kOther[o.type] : "?");
    }
    return ADAS_OK;
}
int adas_engine_detect_sink_supported(const adas_engine* e) {
    if (!e) return 0;
    for (auto& op : e->ops)
        if ((op.f.type == OP_DETECT_V8 || op.f.type == OP_DETECT_V5) && op.det_src[0] >= 0) return 1;
    return 0;
}
int adas_engine_detect_sink_shape(const adas_engine* e, int32_t* layout, int32_t* num_anchors, int32_t* num_classes) {
    ADAS_REQUIRE(e, ADAS_ERR_INVALID, "adas_engine_detect_sink_shape: null engine");
    for (auto& op : e->ops)
        if ((op.f.type == OP_DETECT_V8 || op.f.type == OP_DETECT_V5) && op.det_src[0] >= 0) {
            if (layout) *layout = op.f.type == OP_DETECT_V8 ? ADAS_HEAD_V8 : ADAS_HEAD_V5;
            if (num_classes) *num_classes = (int32_t)op.f.params[0];
            if (num_anchors) *num_anchors = (int32_t)op.f.params[1];
            return ADAS_OK;
        }
    set_error("this engine has no fused Detect kernel");
    return ADAS_ERR_INVALID;
}
int adas_engine_set_detect_sink(adas_engine* e, float* d_best_conf, int32_t* d_best_cls) {
    ADAS_REQUIRE(e && ((d_best_conf == nullptr) == (d_best_cls == nullptr)), ADAS_ERR_INVALID, "adas_engine_set_detect_sink: bad argument");
    ADAS_REQUIRE(!d_best_conf || adas_engine_detect_sink_supported(e), ADAS_ERR_INVALID,
                 "this engine's Detect is not one of the fused kernels (16-bit precisions, the head's last 1x1 convs folded in): it has no per-anchor sink");
    e->sink_conf = d_best_conf;
    e->sink_cls = d_best_cls;
    return ADAS_OK;
}
int adas_engine_layer_info(const adas_engine* e, int layer, char* name, int cap, double* flops, int* kind) {
    ADAS_REQUIRE(e && layer >= 0 && layer < (int)e->ops.size(), ADAS_ERR_INVALID, "bad layer index");
    if (name && cap > 0) snprintf(name, cap, "%s", e->ops[layer].name.c_str());
    if (flops) *flops = e->ops[layer].f.flops;
    if (kind) *kind = (int)e->ops[layer].f.type;
    return ADAS_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------- execution
namespace adas {

int engine_run_op(adas_engine* e, int i, const float* d_in, int batch, hipStream_t st, bool packed_in) {
    EngOp& op = e->ops[i];
    const FileOp& o = op.f;
    unsigned char* wb = (unsigned char*)e->d_weights;
    hipError_t err = hipSuccess;
    if (op.skip) return ADAS_OK;  // folded into the stem launch
    if (o.type == OP_CONV && op.kernel == CONV_STEM) {
        TView cv = make_view(e, o.out_buf, o.out_coff, o.out_c);
        TView pv = cv;
        if (op.fuse_pool >= 0) {
            const FileOp& po = e->ops[op.fuse_pool].f;
            pv = make_view(e, po.out_buf, po.out_coff, po.out_c);
        }
        if (e->prec == PREC_X3 && op.fuse_conv2 >= 0) {
            const EngOp& c2 = e->ops[op.fuse_conv2];
            err = launch_conv_stem2_x3(d_in, batch, e->hdr.in_c, e->hdr.in_h, e->hdr.in_w, o.kh, o.pad, wb + op.w_off, (const float*)(wb + op.b_off), cv,
                                       wb + c2.w_off, (const float*)(wb + c2.b_off), make_view(e, c2.f.out_buf, c2.f.out_coff, c2.f.out_c), st);
        } else if (e->prec == PREC_X3 && op.fuse_pool >= 0) {
            err = launch_conv_stem_pool_x3(d_in, batch, e->hdr.in_c, e->hdr.in_h, e->hdr.in_w, o.pad, wb + op.w_off, (const float*)(wb + op.b_off), cv, pv, st);
        } else if (e->prec == PREC_X3) {
            err = launch_conv_stem_x3(d_in, batch, e->hdr.in_c, e->hdr.in_h, e->hdr.in_w, o.kh, o.pad, o.act, wb + op.w_off, (const float*)(wb + op.b_off), cv, st);
        } else if (op.fuse_conv2 >= 0) {
            const EngOp& c2 = e->ops[op.fuse_conv2];
            err = launch_conv_stem2(d_in, batch, e->hdr.in_c, e->hdr.in_h, e->hdr.in_w, o.kh, o.pad, wb + op.w_off, (const float*)(wb + op.b_off), cv,
                                    wb + c2.w_off, (const float*)(wb + c2.b_off), make_view(e, c2.f.out_buf, c2.f.out_coff, c2.f.out_c), packed_in, e->prec, st);
        } else
            err = launch_conv_stem(d_in, batch, e->hdr.in_c, e->hdr.in_h, e->hdr.in_w, o.kh, o.pad, o.act, wb + op.w_off,
                                   (const float*)(wb + op.b_off), cv, op.fuse_pool >= 0, pv, packed_in, e->prec, st);
        if (err != hipSuccess) {
            set_error("layer %d (%s): stem launch failed: %s", i, op.name.c_str(), hipGetErrorString(err));
            (void)hipGetLastError();
            return ADAS_ERR_HIP;
        }
        return ADAS_OK;
    }
    switch (o.type) {
        case OP_INPUT:
            err = launch_input_nchw(d_in, make_view(e, o.out_buf, 0, 8), batch, e->hdr.in_c, e->prec, st);
            break;
        case OP_CONV: {
            if (op.ds_user >= 0 && ds_folded(e, op.ds_user, batch)) break;   // computed inside the conv it is the shortcut of
            if (op.c2f[0] >= 0) {   // this 1x1 conv, the Bottleneck pair behind it and the block's closing 1x1 conv: one launch
                const EngOp &ca = e->ops[op.c2f[0]], &cb = e->ops[op.c2f[1]], &c2 = e->ops[op.c2f[2]];
                if (e->prec == PREC_X3) {
                    const void* const w4[4] = {wb + op.w_off, wb + ca.w_off, wb + cb.w_off, wb + c2.w_off};
                    const float* const b4[4] = {(const float*)(wb + op.b_off), (const float*)(wb + ca.b_off), (const float*)(wb + cb.b_off), (const float*)(wb + c2.b_off)};
                    err = launch_conv_c2f16_x3(make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), make_view(e, c2.f.out_buf, c2.f.out_coff, c2.f.out_c), w4, b4, batch, st);
                    break;
                }
                err = launch_conv_c2f16(make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), make_view(e, c2.f.out_buf, c2.f.out_coff, c2.f.out_c),
                                        wb + op.w_off, (const float*)(wb + op.b_off), wb + ca.w_off, (const float*)(wb + ca.b_off), wb + cb.w_off,
                                        (const float*)(wb + cb.b_off), wb + c2.w_off, (const float*)(wb + c2.b_off), batch, e->prec, st);
                break;
            }
            if (op.pair_b >= 0) {   // this conv and the one behind it, one launch
                const EngOp& b = e->ops[op.pair_b];
                err = launch_conv_pair(make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), make_view(e, b.f.out_buf, b.f.out_coff, b.f.out_c),
                                       wb + op.w_off, (const float*)(wb + op.b_off), wb + b.w_off, (const float*)(wb + b.b_off), batch,
                                       b.f.res_mode != RES_NONE, e->prec, st);
                break;
            }
            const ConvArgs a = conv_args_of(e, i, batch);
            err = launch_conv(a, e->prec, st);
            break;
        }
        case OP_MAXPOOL:
            if (op.pool3[0] >= 0) {
                const FileOp &q1 = e->ops[op.pool3[0]].f, &q2 = e->ops[op.pool3[1]].f;
                const TView outs[3] = {make_view(e, o.out_buf, o.out_coff, o.out_c), make_view(e, q1.out_buf, q1.out_coff, q1.out_c),
                                       make_view(e, q2.out_buf, q2.out_coff, q2.out_c)};
                err = launch_sppf_pool3(make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), outs, batch, e->prec, st);
                break;
            }
            err = launch_maxpool(make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), make_view(e, o.out_buf, o.out_coff, o.out_c), batch,
                                 o.kh, o.stride, o.pad, e->prec, st);
            break;
        case OP_AVGPOOL:
            err = launch_avgpool(make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), make_view(e, o.out_buf, o.out_coff, o.out_c), batch, o.kh, o.stride,
                                 o.pad, e->prec, st);
            break;
        case OP_DEPTH2SPACE:
            err = launch_depth2space(make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), make_view(e, o.out_buf, o.out_coff, o.out_c), batch, e->prec, st);
            break;
        case OP_DETECT_V6: {
            TView ins[6];
            for (int k = 0; k < 6; ++k) ins[k] = make_view(e, o.in_buf[k], o.in_coff[k], o.in_c[k]);
            int strides[3] = {(int)o.params[2], (int)o.params[3], (int)o.params[4]};
            err = launch_detect_v6(ins, (float*)e->bufs[o.out_buf].d, batch, (int)o.params[0], (int)o.params[1], strides, st);
            break;
        }
        case OP_UPSAMPLE2:
            err = launch_upsample2(make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), make_view(e, o.out_buf, o.out_coff, o.out_c), batch,
                                   e->prec, st);
            break;
        case OP_DETECT_V8: {
            TView ins[6];
            int strides[3] = {(int)o.params[2], (int)o.params[3], (int)o.params[4]};
            if (op.det_src[0] >= 0) {  // decode + the six 1x1 convs in front of it
                const void* wf[6];
                const float* bs[6];
                for (int k = 0; k < 6; ++k) {
                    const EngOp& c = e->ops[op.det_src[k]];
                    ins[k] = make_view(e, c.f.in_buf[0], c.f.in_coff[0], c.f.in_c[0]);
                    wf[k] = wb + c.w_off;
                    bs[k] = (const float*)(wb + c.b_off);
                }
                if (e->prec == PREC_X3)
                    err = launch_detect_v8_fused_x3(ins, wf, bs, e->ops[op.det_src[0]].kpad / 32, e->ops[op.det_src[1]].kpad / 32, (float*)e->bufs[o.out_buf].d, batch,
                                                    (int)o.params[0], (int)o.params[1], strides, st, e->sink_conf, e->sink_cls);
                else
                    err = launch_detect_v8_fused(ins, wf, bs, (float*)e->bufs[o.out_buf].d, batch, (int)o.params[0], (int)o.params[1], strides, e->prec, st, e->sink_conf, e->sink_cls);
                break;
            }
            for (int k = 0; k < 6; ++k) ins[k] = make_view(e, o.in_buf[k], o.in_coff[k], o.in_c[k]);
            err = launch_detect_v8(ins, (float*)e->bufs[o.out_buf].d, batch, (int)o.params[0], (int)o.params[1], strides, st);
            break;
        }
        case OP_DETECT_V5: {
            TView ins[3];
            int strides[3] = {(int)o.params[2], (int)o.params[3], (int)o.params[4]};
            if (op.det_src[0] >= 0) {  // decode + the three 1x1 convs in front of it
                const void* wf[3];
                const float* bs[3];
                for (int k = 0; k < 3; ++k) {
                    const EngOp& c = e->ops[op.det_src[k]];
                    ins[k] = make_view(e, c.f.in_buf[0], c.f.in_coff[0], c.f.in_c[0]);
                    wf[k] = wb + c.w_off;
                    bs[k] = (const float*)(wb + c.b_off);
                }
                err = launch_detect_v5_fused(ins, wf, bs, (float*)e->bufs[o.out_buf].d, batch, (int)o.params[0], (int)o.params[1], strides,
                                             (const float*)(wb + op.w_off), e->prec, st, e->sink_conf, e->sink_cls);
                break;
            }
            for (int k = 0; k < 3; ++k) ins[k] = make_view(e, o.in_buf[k], o.in_coff[k], o.in_c[k]);
            err = launch_detect_v5(ins, (float*)e->bufs[o.out_buf].d, batch, (int)o.params[0], (int)o.params[1], strides,
                                   (const float*)(wb + op.w_off), st);
            break;
        }
        case OP_DWCONV: {
            TView r{};
            if (o.res_mode != RES_NONE) r = make_view(e, o.res_buf, o.res_coff, o.out_c);
            err = launch_dwconv(make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), make_view(e, o.out_buf, o.out_coff, o.out_c), r, (int)o.res_mode,
                                (const float*)(wb + op.w_off), (const float*)(wb + op.b_off), batch, (int)o.kh, (int)o.stride, (int)o.pad, (int)o.act,
                                e->prec, st);
            break;
        }
        case OP_SE_GATE: {
            const float* w1 = (const float*)(wb + op.w_off);
            const float* w2 = (const float*)(wb + op.b_off);
            TView scratch{};
            const bool has_scratch = o.res_buf >= 0 && o.res_buf < (int32_t)e->bufs.size();   // res_buf: the per-frame scratch of the two-launch form
            if (has_scratch) scratch = make_view(e, o.res_buf, 0, e->bufs[o.res_buf].c);
            err = launch_se_gate(make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), make_view(e, o.out_buf, o.out_coff, o.out_c), w1, w2, (int)o.params[0],
                                 (int)o.params[1], (int)o.params[2], batch, e->prec, st, has_scratch ? &scratch : nullptr);
            break;
        }
        case OP_SCALE:
            err = launch_scale(make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), make_view(e, o.in_buf[1], o.in_coff[1], o.in_c[1]),
                               make_view(e, o.out_buf, o.out_coff, o.out_c), batch, e->prec, st);
            break;
        case OP_SHUFFLE:
            err = launch_shuffle(make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), make_view(e, o.out_buf, o.out_coff, o.out_c), (int)o.params[0], batch, e->prec, st);
            break;
        case OP_WSUM: {
            TView ins[3];
            for (uint32_t k = 0; k < o.n_in && k < 3; ++k) ins[k] = make_view(e, o.in_buf[k], o.in_coff[k], o.in_c[k]);
            err = launch_wsum((int)o.n_in, ins, o.params, make_view(e, o.out_buf, o.out_coff, o.out_c), batch, (int)o.act, e->prec, st);
            break;
        }
        case OP_ATTENTION:
            err = launch_attention(make_view(e, o.in_buf[0], o.in_coff[0], o.in_c[0]), make_view(e, o.out_buf, o.out_coff, o.out_c), batch,
                                   (int)o.params[0], (int)o.params[1], (int)o.params[2], o.params[3], e->prec, st);
            break;
        case OP_LAYERNORM: {
            const EngBuf& ib = e->bufs[o.in_buf[0]];
            int len = ib.h * ib.w * ib.c;
            if (!ib.f32) { set_error("layernorm input must be fp32"); return ADAS_ERR_FORMAT; }
            err = launch_layernorm((const float*)ib.d, e->bufs[o.out_buf].d, (const float*)(wb + op.w_off), (const float*)(wb + op.b_off),
                                   batch, len, o.params[0], e->prec, st);
            break;
        }
        default:
            set_error("unknown op type %u in layer %d (%s)", o.type, i, op.name.c_str());
            return ADAS_ERR_FORMAT;
    }
    if (err != hipSuccess) {
        set_error("layer %d (%s): launch failed: %s", i, op.name.c_str(), hipGetErrorString(err));
        (void)hipGetLastError();
        return ADAS_ERR_HIP;
    }
    return ADAS_OK;
}

// Builds the multi-layer launches of this batch size (device tables: allocations and copies, so never inside a stream capture --
// adas_pipeline_* prepares before it captures; a forward that finds nothing prepared while capturing runs per-layer launches).
static void prepare_groups(adas_engine* e, int batch) {
    std::vector<GroupRun>& runs = e->groups[batch];
    const int n = (int)e->ops.size();
    int i = 0;
    while (i < n) {
        std::vector<ConvArgs> layers;
        std::vector<int> ops;
        int j = i, last = i;
        for (; j < n; ++j) {
            if (e->ops[j].skip) continue;
            if (e->ops[j].f.type == OP_CONV && e->ops[j].ds_user >= 0 && ds_folded(e, e->ops[j].ds_user, batch)) continue;
            ConvArgs a;
            if (!group_candidate(e, j, batch, &a) || (int)layers.size() >= ML_MAX_LAYERS) break;
            layers.push_back(a); ops.push_back(j);
            last = j;
        }
        if (layers.size() < 2) { i = (layers.empty() ? j : last) + 1; continue; }
        const std::vector<int> level = ml_levels(layers);
        int nlev = 0;
        for (int v : level) nlev = v + 1 > nlev ? v + 1 : nlev;
        GroupRun run;
        run.first = ops.front(); run.last = last; run.ops = ops;
        bool any_group = false, ok = true;
        for (int lv = 0; lv < nlev && ok; ++lv) {
            std::vector<int> idx;
            for (size_t k = 0; k < layers.size(); ++k)
                if (level[k] == lv) idx.push_back((int)k);
            for (size_t c0 = 0; c0 < idx.size() && ok; c0 += ML_GROUP_MAX) {   // at most ML_GROUP_MAX layers per launch
                const size_t c1 = c0 + ML_GROUP_MAX < idx.size() ? c0 + ML_GROUP_MAX : idx.size();
                GroupStep st;
                if (c1 - c0 >= 2) {
                    std::vector<ConvArgs> sub;
                    for (size_t k = c0; k < c1; ++k) { sub.push_back(layers[idx[k]]); st.members.push_back(ops[idx[k]]); }
                    std::string why;
                    st.group = ml_group_create(sub, e->prec, &why);
                    ok = st.group != nullptr;
                    any_group = true;
                } else {
                    st.op = ops[idx[c0]];
                    st.members.push_back(st.op);
                }
                if (ok) run.steps.push_back(st);
            }
        }
        if (ok && any_group) runs.push_back(run);
        else
            for (auto& st : run.steps) ml_group_destroy(st.group);
        i = last + 1;
    }
}

// Tables are kept for the life of the engine (a captured hipGraph may reference them): at most this many distinct batch sizes get
// them, later ones run one launch per layer.
constexpr size_t kMaxPreparedBatches = 16;

// The allocations and copies below must not land in -- or invalidate -- a stream capture the calling thread has open on ANOTHER stream
// (engine_forward only knows its own): they run with the thread's capture mode relaxed, restored on every exit.
struct RelaxedCaptureMode {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    bool ok;
    RelaxedCaptureMode() { ok = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess; if (!ok) (void)hipGetLastError(); }
    ~RelaxedCaptureMode() { if (ok && hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) (void)hipGetLastError(); }
};

int engine_prepare(adas_engine* e, int batch) {
    const bool want_groups = e->group_on && !e->groups.count(batch), want_ml = ml_enabled(e) && !e->ml.count(batch);
    if (!want_groups && !want_ml) return ADAS_OK;
    RelaxedCaptureMode relaxed;
    if (want_groups) {
        if (e->groups.size() >= kMaxPreparedBatches) e->groups[batch];   // an empty list: per-layer launches for this batch size
        else prepare_groups(e, batch);
    }
    if (!want_ml) return ADAS_OK;
    if (e->ml.size() >= kMaxPreparedBatches) { e->ml[batch]; return ADAS_OK; }
    std::vector<MlSeg>& segs = e->ml[batch];
    static int min_layers = -1, max_items = -1;
    if (min_layers < 0) { const char* v = getenv("ADAS_ML_MIN_LAYERS"); min_layers = v ? atoi(v) : 2; if (min_layers < 1) min_layers = 1; }
    if (max_items < 0) { const char* v = getenv("ADAS_ML_MAX_LAYER_ITEMS"); max_items = v ? atoi(v) : 0; }   // experiments: keep layers with more items out
    const int n = (int)e->ops.size();
    int i = 0;
    while (i < n) {
        std::vector<ConvArgs> layers;
        std::vector<int> kernels, ops;
        int j = i, last = i;
        for (; j < n; ++j) {
            if (e->ops[j].skip) continue;            // launches nothing (fused into a neighbour): transparent
            if (e->ops[j].f.type == OP_CONV && e->ops[j].ds_user >= 0 && ds_folded(e, e->ops[j].ds_user, batch)) continue;
            ConvArgs a;
            if (!ml_candidate(e, j, batch, &a) || (int)layers.size() >= ML_MAX_LAYERS) break;
            if (max_items > 0) {
                const long wgs = (long)((a.m + 255) / 256) * ((a.out.c + 63) / 64);
                if (wgs > max_items) break;
            }
            layers.push_back(a); kernels.push_back(e->ops[j].kernel); ops.push_back(j);
            last = j;
        }
        if ((int)layers.size() >= min_layers) {
            std::string why;
            MlPlanInfo info;
            MlPlan* pl = ml_plan_create(layers, kernels, e->prec, &why, &info);
            if (pl) {
                MlSeg sg;
                sg.first = ops.front(); sg.last = last; sg.n_layers = (int)layers.size(); sg.n_items = info.n_items; sg.plan = pl; sg.ops = ops;
                segs.push_back(sg);
            }
            i = last + 1;
        } else {
            i = (layers.empty() ? j : last) + 1;
        }
    }
    return ADAS_OK;
}

static bool stream_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}

int engine_forward(adas_engine* e, const float* d_in, int batch, hipStream_t st, bool packed_in) {
    if (((ml_enabled(e) && !e->ml.count(batch)) || (e->group_on && !e->groups.count(batch))) && !stream_capturing(st)) {
        int rc = engine_prepare(e, batch);
        if (rc != ADAS_OK) return rc;
    }
    const std::vector<MlSeg>* segs = ml_segments(e, batch);
    const std::vector<GroupRun>* runs = group_runs(e, batch);
    size_t si = 0, gi = 0;
    for (int i = 0; i < (int)e->ops.size(); ++i) {
        if (runs && gi < runs->size() && (*runs)[gi].first == i) {   // a run of halo convs, level by level: independent layers share a launch
            const GroupRun& run = (*runs)[gi++];
            for (auto& step : run.steps) {
                if (step.group) {
                    hipError_t err = ml_group_launch(step.group, st);
                    if (err != hipSuccess) {
                        set_error("layers %d..%d: grouped launch failed: %s", run.first, run.last, hipGetErrorString(err));
                        (void)hipGetLastError();
                        return ADAS_ERR_HIP;
                    }
                } else {
                    int rc = engine_run_op(e, step.op, d_in, batch, st, packed_in);
                    if (rc != ADAS_OK) return rc;
                }
            }
            i = run.last;
            continue;
        }
        if (segs && si < segs->size() && (*segs)[si].first == i) {
            const MlSeg& sg = (*segs)[si++];
            hipError_t err = ml_launch(sg.plan, st);
            if (err != hipSuccess) {
                set_error("layers %d..%d (%s ...): multi-layer launch failed: %s", sg.first, sg.last, e->ops[sg.first].name.c_str(), hipGetErrorString(err));
                (void)hipGetLastError();
                return ADAS_ERR_HIP;
            }
            i = sg.last;
            continue;
        }
        int rc = engine_run_op(e, i, d_in, batch, st, packed_in);
        if (rc != ADAS_OK) return rc;
    }
    return ADAS_OK;
}
}  // namespace adas

extern "C" {

int adas_engine_infer_device(adas_engine* e, const float* d_input, int batch, void* stream) {
    ADAS_REQUIRE(e && d_input && batch > 0 && batch <= e->max_batch, ADAS_ERR_INVALID, "adas_engine_infer_device: bad argument (batch %d, max %d)", batch,
                 e ? e->max_batch : 0);
    e->last = (hipStream_t)stream;
    return engine_forward(e, d_input, batch, (hipStream_t)stream);
}

int adas_engine_prepare(adas_engine* e, int batch) {
    ADAS_REQUIRE(e && batch > 0 && batch <= e->max_batch, ADAS_ERR_INVALID, "adas_engine_prepare: bad argument (batch %d, max %d)", batch, e ? e->max_batch : 0);
    return engine_prepare(e, batch);
}
int adas_engine_ml_info(const adas_engine* e, int batch, int32_t* n_launches, int32_t* n_layers, int32_t* n_items) {
    ADAS_REQUIRE(e && batch > 0, ADAS_ERR_INVALID, "adas_engine_ml_info: bad argument");
    int nl = 0, ni = 0, ns = 0;
    if (const std::vector<MlSeg>* segs = ml_segments(e, batch))
        for (auto& sg : *segs) { ++ns; nl += sg.n_layers; ni += sg.n_items; }
    if (n_launches) *n_launches = ns;
    if (n_layers) *n_layers = nl;
    if (n_items) *n_items = ni;
    return ADAS_OK;
}
int adas_engine_ml_status(const adas_engine* e, int batch, uint32_t* error_word) {
    ADAS_REQUIRE(e && error_word, ADAS_ERR_INVALID, "adas_engine_ml_status: bad argument");
    *error_word = 0;
    if (const std::vector<MlSeg>* segs = ml_segments(e, batch))
        for (auto& sg : *segs) {
            unsigned w = 0;
            ADAS_REQUIRE(ml_plan_status(sg.plan, &w) == 0, ADAS_ERR_HIP, "adas_engine_ml_status: could not read the control block");
            if (w) {
                *error_word = w;
                set_error("multi-layer launch of layers %d..%d: a dependency wait timed out (item %u)", sg.first, sg.last, w - 1);
                return ADAS_ERR_HIP;
            }
        }
    return ADAS_OK;
}
int adas_engine_ml_counters(const adas_engine* e, int batch, int launch, uint32_t head16[16]) {
    ADAS_REQUIRE(e && head16, ADAS_ERR_INVALID, "adas_engine_ml_counters: bad argument");
    const std::vector<MlSeg>* segs = ml_segments(e, batch);
    ADAS_REQUIRE(segs && launch >= 0 && launch < (int)segs->size(), ADAS_ERR_INVALID, "adas_engine_ml_counters: no multi-layer launch %d at batch %d", launch, batch);
    unsigned w = 0;
    ADAS_REQUIRE(ml_plan_status((*segs)[launch].plan, &w, head16) == 0, ADAS_ERR_HIP, "adas_engine_ml_counters: could not read the control block");
    return ADAS_OK;
}
int adas_engine_launch_count(adas_engine* e, int batch) {
    if (!e || batch <= 0 || batch > e->max_batch) return -1;
    const std::vector<MlSeg>* segs = ml_segments(e, batch);
    const std::vector<GroupRun>* runs = group_runs(e, batch);
    size_t si = 0, gi = 0;
    int n = 0;
    for (int i = 0; i < (int)e->ops.size(); ++i) {
        if (runs && gi < runs->size() && (*runs)[gi].first == i) { n += (int)(*runs)[gi].steps.size(); i = (*runs)[gi++].last; continue; }
        if (segs && si < segs->size() && (*segs)[si].first == i) { ++n; i = (*segs)[si++].last; continue; }
        const EngOp& op = e->ops[i];
        if (op.skip) continue;
        if (op.f.type == OP_CONV && op.ds_user >= 0 && ds_folded(e, op.ds_user, batch)) continue;
        ++n;
    }
    return n;
}

int adas_debug_ml_plan(const adas_ml_layer_desc* layers, int n_layers, int batch, int precision, int32_t* deps, int32_t* targets, uint64_t* items,
                       int items_cap, int32_t summary[4]) {
    ADAS_REQUIRE(layers && n_layers > 0 && batch > 0 && deps && targets && summary, ADAS_ERR_INVALID, "adas_debug_ml_plan: bad argument");
    std::vector<ConvArgs> ls;
    std::vector<int> ks;
    auto view = [](const adas_ml_view& v) {
        TView t;
        t.p = (void*)(uintptr_t)v.buf; t.cs = v.cs; t.coff = v.coff; t.c = v.c; t.h = v.h; t.w = v.w; t.f32 = 0;
        return t;
    };
    for (int i = 0; i < n_layers; ++i) {
        const adas_ml_layer_desc& d = layers[i];
        ConvArgs a;
        a.in = view(d.x); a.out = view(d.y);
        if (d.res_mode != RES_NONE) a.res = view(d.res);
        else { a.res = a.out; a.res.p = nullptr; }
        a.wgt = (const void*)(uintptr_t)0x1000; a.bias = (const float*)(uintptr_t)0x1000;
        a.n = batch; a.kh = a.kw = d.kernel == CONV_PW ? 1 : 3; a.stride = d.stride; a.pad = d.kernel == CONV_PW ? 0 : 1; a.act = d.act; a.res_mode = d.res_mode;
        a.k = a.kh * a.kw * a.in.c; a.kpad = (a.kh * a.kw * ((a.in.c + 31) / 32 * 32) + 31) / 32 * 32; a.m = batch * a.out.h * a.out.w; a.max_n = batch; a.prec = precision;
        a.halo_bn = d.halo_bn;
        if (d.up_c > 0) { a.up = view(d.up); a.up_c = d.up_c; }
        ls.push_back(a); ks.push_back(d.kernel);
    }
    std::string why;
    MlPlanInfo info;
    MlPlan* pl = ml_plan_create(ls, ks, precision, &why, &info, true);
    ADAS_REQUIRE(pl, ADAS_ERR_INVALID, "adas_debug_ml_plan: %s", why.c_str());
    ml_plan_destroy(pl);
    for (int i = 0; i < n_layers; ++i)
        for (int k = 0; k < ML_MAX_DEPS; ++k) {
            deps[i * ML_MAX_DEPS + k] = k < (int)info.deps[i].size() ? info.deps[i][k] : -1;
            targets[i * ML_MAX_DEPS + k] = k < (int)info.targets[i].size() ? info.targets[i][k] : 0;
        }
    summary[0] = info.n_items; summary[1] = info.grid; summary[2] = (int32_t)info.lds; summary[3] = info.order;
    if (items) {
        ADAS_REQUIRE(items_cap >= info.n_items, ADAS_ERR_CAPACITY, "adas_debug_ml_plan: %d items, room for %d", info.n_items, items_cap);
        for (int k = 0; k < info.n_items; ++k) items[k] = info.item_words[k];
    }
    return ADAS_OK;
}

int adas_engine_precision(const adas_engine* e) { return e ? e->prec : -1; }
int adas_engine_model_io_half(const adas_engine* e) { return e ? (int)((e->hdr.in_cpad >> 16) & 1u) : 0; }

int adas_engine_accepts_packed_input(const adas_engine* e) {
    // (the split precision's stem reads the fp32 seam tensor: a 16-bit packed pixel could not carry its 22 bits)
    return (e && e->prec != PREC_X3 && e->ops.size() >= 2 && e->ops[0].skip && e->ops[1].kernel == CONV_STEM) ? 1 : 0;
}

int adas_engine_infer_device_packed(adas_engine* e, const uint16_t* d_input_nhwc4, int batch, void* stream) {
    ADAS_REQUIRE(e && d_input_nhwc4 && batch > 0 && batch <= e->max_batch, ADAS_ERR_INVALID, "adas_engine_infer_device_packed: bad argument");
    ADAS_REQUIRE(adas_engine_accepts_packed_input(e), ADAS_ERR_INVALID,
                 "this engine's first layer is not the fused stem (fp32 mode or ADAS_NO_STEM): feed the fp32 NCHW tensor instead");
    e->last = (hipStream_t)stream;
    return engine_forward(e, reinterpret_cast<const float*>(d_input_nhwc4), batch, (hipStream_t)stream, true);
}

const float* adas_engine_output_device(const adas_engine* e, int index) {
    if (!e || index < 0 || index >= (int)e->outs.size()) return nullptr;
    const EngOut& o = e->outs[index];
    return (const float*)e->bufs[o.buf].d + o.offset;
}

int adas_engine_infer_host(adas_engine* e, const float* h_input, int batch, float* const* h_outputs) {
    ADAS_REQUIRE(e && h_input && h_outputs && batch > 0 && batch <= e->max_batch, ADAS_ERR_INVALID, "adas_engine_infer_host: bad argument");
    size_t in_bytes = (size_t)batch * e->hdr.in_c * e->hdr.in_h * e->hdr.in_w * 4;
    ADAS_HIP_TRY(hipMemcpyAsync(e->d_input, h_input, in_bytes, hipMemcpyHostToDevice, 0));
    int rc = adas_engine_infer_device(e, e->d_input, batch, nullptr);
    if (rc != ADAS_OK) return rc;
    for (size_t i = 0; i < e->outs.size(); ++i) {
        const EngOut& o = e->outs[i];
        const EngBuf& b = e->bufs[o.buf];
        size_t frame_stride = (size_t)b.h * b.w * b.c;  // floats per frame in the backing buffer
        ADAS_HIP_TRY(hipMemcpy2DAsync(h_outputs[i], o.elems * 4, (const float*)b.d + o.offset, frame_stride * 4, o.elems * 4, batch,
                                      hipMemcpyDeviceToHost, 0));
    }
    ADAS_HIP_TRY(hipStreamSynchronize(0));
    // opt-in multi-layer launches (ADAS_ML=1): a dependency wait that timed out leaves its item uncomputed -- the caller gets an error,
    // never the partial outputs
    if (ml_enabled(e)) {
        uint32_t w = 0;
        int rc = adas_engine_ml_status(e, batch, &w);
        if (rc != ADAS_OK) return rc;
    }
    return ADAS_OK;
}

int adas_engine_profile(adas_engine* e, const float* d_input, int batch, int iters, float* ms_per_layer, int max_layers, int* num_layers) {
    ADAS_REQUIRE(e && d_input && batch > 0 && batch <= e->max_batch && iters > 0 && ms_per_layer, ADAS_ERR_INVALID, "adas_engine_profile: bad argument");
    const int n = (int)e->ops.size();
    ADAS_REQUIRE(max_layers >= n, ADAS_ERR_INVALID, "need room for %d layers", n);
    if ((int)e->events.size() < n + 1) {
        e->events.resize(n + 1, nullptr);
        for (auto& ev : e->events)
            if (!ev) ADAS_HIP_TRY(hipEventCreate(&ev));
    }
    for (int i = 0; i < n; ++i) ms_per_layer[i] = 0.f;
    // an event record is itself a packet on the stream (~5 us between two records with nothing in between): measured here and taken
    // off every layer, so that a layer that launches nothing (fused / folded into a neighbour) reads 0 and the per-layer sum is the
    // kernels' time, not kernels + markers
    float marker_ms = 0.f;
    {
        const int reps = n < 8 ? n : 8;
        for (int r = 0; r <= reps; ++r) ADAS_HIP_TRY(hipEventRecord(e->events[r], 0));
        ADAS_HIP_TRY(hipStreamSynchronize(0));
        float lo = 1e30f;
        for (int r = 1; r < reps; ++r) {   // (the first gap carries the stream's wake-up)
            float ms = 0.f;
            ADAS_HIP_TRY(hipEventElapsedTime(&ms, e->events[r], e->events[r + 1]));
            lo = ms < lo ? ms : lo;
        }
        marker_ms = lo < 1e29f ? lo : 0.f;
    }
    {
        int rc = engine_prepare(e, batch);
        if (rc != ADAS_OK) return rc;
    }
    const std::vector<MlSeg>* segs = ml_segments(e, batch);
    const std::vector<GroupRun>* runs = group_runs(e, batch);
    struct StepMark { int layer, ev, prev; };   // prev: index into step_events, or -(layer index + 1) of the layer event that opens the run
    for (int it = 0; it < iters; ++it) {
        ADAS_HIP_TRY(hipEventRecord(e->events[0], 0));
        size_t si = 0, gi = 0, n_step_ev = 0;
        std::vector<StepMark> step_marks;
        for (int i = 0; i < n; ++i) {
            if (runs && gi < runs->size() && (*runs)[gi].first == i) {
                // a run of halo convs, launched level by level: one event per STEP, a step's time goes to its first member (the layer
                // adas_engine_layer_kernel labels with the step's kernel), every other layer of the run reads 0
                const GroupRun& run = (*runs)[gi++];
                const size_t base = step_marks.size();
                for (auto& step : run.steps) {
                    if (step.group) {
                        hipError_t err = ml_group_launch(step.group, 0);
                        if (err != hipSuccess) return hip_fail(err, "grouped launch", __FILE__, __LINE__);
                    } else {
                        int rc = engine_run_op(e, step.op, d_input, batch, 0);
                        if (rc != ADAS_OK) return rc;
                    }
                    if (n_step_ev >= e->step_events.size()) {
                        hipEvent_t ev = nullptr;
                        ADAS_HIP_TRY(hipEventCreate(&ev));
                        e->step_events.push_back(ev);
                    }
                    ADAS_HIP_TRY(hipEventRecord(e->step_events[n_step_ev], 0));
                    step_marks.push_back({step.members.front(), (int)n_step_ev, step_marks.size() == base ? -(i + 1) : (int)n_step_ev - 1});
                    ++n_step_ev;
                }
                for (int k = i; k <= run.last; ++k) ADAS_HIP_TRY(hipEventRecord(e->events[k + 1], 0));
                i = run.last;
                continue;
            }
            if (segs && si < segs->size() && (*segs)[si].first == i) {   // a multi-layer launch: its time goes to its first layer, the rest read 0
                const MlSeg& sg = (*segs)[si++];
                hipError_t err = ml_launch(sg.plan, 0);
                if (err != hipSuccess) return hip_fail(err, "multi-layer launch", __FILE__, __LINE__);
                for (int k = i; k <= sg.last; ++k) ADAS_HIP_TRY(hipEventRecord(e->events[k + 1], 0));
                i = sg.last;
                continue;
            }
            int rc = engine_run_op(e, i, d_input, batch, 0);
            if (rc != ADAS_OK) return rc;
            ADAS_HIP_TRY(hipEventRecord(e->events[i + 1], 0));
        }
        ADAS_HIP_TRY(hipStreamSynchronize(0));
        std::vector<char> in_run(n, 0);
        if (runs)
            for (auto& run : *runs)
                for (int k = run.first; k <= run.last; ++k) in_run[k] = 1;
        for (int i = 0; i < n; ++i) {
            if (in_run[i]) continue;          // layers of a grouped run are timed per step below
            float ms = 0.f;
            ADAS_HIP_TRY(hipEventElapsedTime(&ms, e->events[i], e->events[i + 1]));
            ms -= marker_ms;
            ms_per_layer[i] += (ms > 0.f ? ms : 0.f) / (float)iters;
        }
        for (auto& mk : step_marks) {
            float ms = 0.f;
            hipEvent_t from = mk.prev < 0 ? e->events[-mk.prev - 1] : e->step_events[mk.prev];
            ADAS_HIP_TRY(hipEventElapsedTime(&ms, from, e->step_events[mk.ev]));
            ms -= marker_ms;
            ms_per_layer[mk.layer] += (ms > 0.f ? ms : 0.f) / (float)iters;
        }
    }
    if (num_layers) *num_layers = n;
    return ADAS_OK;
}

int adas_engine_fetch_activation(adas_engine* e, int layer, int batch, float* h_out, int64_t dims[4]) {
    ADAS_REQUIRE(e && layer >= 0 && layer < (int)e->ops.size() && batch > 0 && batch <= e->max_batch, ADAS_ERR_INVALID, "bad layer/batch");
    const FileOp& o = e->ops[layer].f;
    ADAS_REQUIRE(!(e->ops[layer].skip && o.type == OP_CONV && (e->ops[layer].kernel == CONV_PW || e->ops[layer].kernel == CONV_DET5)), ADAS_ERR_INVALID,
                 "layer %d (%s) is fused into the Detect launch and has no materialised activation (ADAS_NO_DETECT_FUSE=1 keeps it)", layer,
                 e->ops[layer].name.c_str());
    ADAS_REQUIRE(!(e->ops[layer].skip && o.type == OP_INPUT) &&
                     !(e->ops[layer].kernel == CONV_STEM && (e->ops[layer].fuse_pool >= 0 || e->ops[layer].fuse_conv2 >= 0)), ADAS_ERR_INVALID,
                 "layer %d (%s) is fused into the stem launch and has no materialised activation (ADAS_NO_STEM=1 keeps it)", layer,
                 e->ops[layer].name.c_str());
    ADAS_REQUIRE(!(e->ops[layer].ds_user >= 0 && ds_folded(e, e->ops[layer].ds_user, batch)), ADAS_ERR_INVALID,
                 "layer %d (%s) is a projection shortcut computed inside the conv that adds it at this batch (ADAS_NO_DS_FUSE=1 keeps it)", layer,
                 e->ops[layer].name.c_str());
    ADAS_REQUIRE(!(e->ops[layer].skip && o.type == OP_UPSAMPLE2), ADAS_ERR_INVALID,
                 "layer %d (%s) is folded into its consumer's loads and has no materialised activation (ADAS_NO_UPSAMPLE_FOLD=1 keeps it)", layer,
                 e->ops[layer].name.c_str());
    {   // a fused C2f launch materialises only its cv2 output: cv1 and the Bottleneck's two convs stay in LDS
        const bool c2f_hidden = e->ops[layer].c2f[0] >= 0 || (in_c2f(e, layer) && !is_c2f_tail(e, layer));
        ADAS_REQUIRE(!c2f_hidden, ADAS_ERR_INVALID,
                     "layer %d (%s) is computed inside a fused C2f launch: its activation stays in LDS (ADAS_NO_C2F_FUSE=1 keeps it)", layer,
                     e->ops[layer].name.c_str());
    }
    ADAS_REQUIRE(e->ops[layer].pair_b < 0, ADAS_ERR_INVALID,
                 "layer %d (%s) is the first conv of a fused 3x3 pair: its activation stays in LDS (ADAS_NO_PAIR_FUSE=1 keeps it)", layer,
                 e->ops[layer].name.c_str());
    TView v = make_view(e, o.out_buf, o.out_coff, o.out_c);
    if (o.type == OP_INPUT) v.c = 8;
    if (dims) { dims[0] = batch; dims[1] = v.c; dims[2] = v.h; dims[3] = v.w; }
    if (!h_out) return ADAS_OK;
    size_t n = (size_t)batch * v.c * v.h * v.w;
    float* d_tmp = nullptr;
    ADAS_HIP_TRY(hipMalloc((void**)&d_tmp, n * 4));
    hipError_t err = launch_nhwc_to_nchw(v, d_tmp, batch, e->prec, 0);
    if (err == hipSuccess) err = hipMemcpy(h_out, d_tmp, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_tmp);
    if (err != hipSuccess) return hip_fail(err, "fetch_activation", __FILE__, __LINE__);
    return ADAS_OK;
}

}  // extern "C"
