// conv_ml.hip -- multi-layer persistent convolution launch (round 5; interface and motivation: conv_ml.h).
//
// DEVICE SIDE.  grid = 2 workgroups per CU (256 threads, <= 80 KB LDS each), all of them running conv_ml_kernel's loop:
//   ticket  t = atomicAdd(ctl[0]), requested by thread 0 while the PREVIOUS item runs; as soon as it is back (after the tile body) thread 0
//           also fetches the item's 64-byte record, under the store drain -- the next iteration starts with ticket and record in registers
//   record  {tile | chunk, layer, channel block, frame, kind, the <= 6 arrival counters to wait for and their targets}: broadcast through LDS
//   wait    lane k < n_dep polls counter k (relaxed agent-scope loads + s_sleep) while the layer's descriptor comes in through the scalar
//           cache; every wait is bounded: a timeout raises ctl[1], the item is abandoned, every workgroup drains -- never a hang
//   run     the SAME tile body the per-layer kernel runs (conv_halo_body.h halo_tile, or the pointwise tile below) with ML = true
//   publish every wave `s_waitcnt vmcnt(0)`, barrier, lane 0: atomicAdd(ctl[16 + layer * F + frame], 1)
// Visibility between workgroups of one launch (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility";
// cdna_hip_programming.md 6 G16): a CU's L1 is never refreshed by other CUs' stores and the eight XCD L2s are not coherent.  The form
// used here is the guide's "sc1 stores AND sc1 loads": every activation store of an ML tile is write-through (buffer_store ... sc1),
// every activation load bypasses L1 (buffer_load ... sc1), the counter is touched with agent-scope atomics only, and the storing waves
// drain (vmcnt(0)) before the one arrival.  Weights and biases are immutable and keep the cached path.
// Deadlock freedom: the item table is ordered so that every item's producers (all items of the producer layers in the same frame) have
// SMALLER tickets (verified on the host for every table); a ticket is only ever held by a resident workgroup; so the smallest unfinished
// ticket is always runnable.  No assumption about dispatch order, residency or XCD placement.
//
// HOST SIDE.  ml_plan_create derives, from the layers' views alone: read-after-write edges (a layer reads a channel range some earlier layer
// of the launch writes), write-after-read and write-after-write edges (buffer reuse), reduces them transitively (frame-complete is
// transitive), enumerates the items and orders them with a list scheduler (critical path first, independent branches as filler).
#include "conv_halo_body.h"
#include "conv_ml.h"
#include <algorithm>
#include <queue>
#include <stdlib.h>
#include <string.h>

namespace adas {

enum { MLK_H64_S1_256 = 0, MLK_H64_S1_128, MLK_H48_S1_256, MLK_H48_S1_128, MLK_H64_S2, MLK_PW,
       MLK_H32_S1_128, MLK_H16_S1_128 /* grouped launch only: the narrow channel blocks of batch-1 engines (plan_halo_bn) */, MLK_NONE = -1 };

struct MlPwDev {
    const uint16_t* in;
    const uint16_t* wfrag;   // [NT][KS][64][8] MFMA-fragment order (CONV_PW / CONV_FC packing)
    const float* bias;
    uint16_t* out;
    const uint16_t* up;      // half-resolution source of the first up_ks K steps (nearest 2x upsample folded in), or null
    int in_cs, in_coff, cin, out_cs, out_coff, cout;
    int HW, W, P, chunks;    // pixels per frame, row width, pixels per item, items per frame and channel block
    int KS, NT, NTL, act;    // K steps, feature tiles, feature tiles per channel block
    int up_cs, up_coff, up_ks, up_W, up_HW, pad0;
};

struct MlLayerDev {
    int kind, n_dep, per_img, pad;
    int dep_row[ML_MAX_DEPS], dep_target[ML_MAX_DEPS];
    union U {
        HaloDev h;
        MlPwDev p;
    } u;
};
static_assert(sizeof(MlLayerDev) % 8 == 0, "MlLayerDev layout");

// One work item as the kernel reads it: everything up to the dependency wait in ONE 64-byte line.
struct MlItemRec {
    uint32_t tile;                    // halo: tile index (frame * tiles per frame + t); pointwise: chunk of the frame
    uint32_t where;                   // layer | channel block << 8 | frame << 16
    uint32_t kind;                    // MLK_* | n_dep << 8
    uint32_t arrive;                  // index (into ctl) of the counter this item increments: 16 + layer * frames + frame
    uint32_t dep_idx[ML_MAX_DEPS];    // counters to wait for
    uint32_t dep_target[ML_MAX_DEPS];
};
static_assert(sizeof(MlItemRec) == 64, "MlItemRec layout");

struct MlArgs {
    const MlLayerDev* layers;
    const MlItemRec* items;
    unsigned* ctl;           // [0] ticket, [1] error, [16 + layer * frames + frame] arrivals
    int n_items, frames, spin_limit, rec_off;   // rec_off: byte offset of the record slot in dynamic LDS (behind the largest tile scratch)
};
constexpr int ML_CTL_HEAD = 16;

// immutable tables: read through the constant address space so that a uniform address gives scalar loads whatever the kernel stores elsewhere
template <typename T>
__device__ __forceinline__ void ml_copy_const(T& dst, const void* src) {
    static_assert(sizeof(T) % 4 == 0, "dword copies");
    constexpr int N = sizeof(T) / 4;
    typedef const uint32_t __attribute__((address_space(4))) * cptr;
    cptr s = (cptr)(uintptr_t)src;
    uint32_t w[N];
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = s[i];
    // pin: the values exist in SGPRs HERE (the loads are in flight from this point, e.g. under a dependency wait) -- left alone, hipcc sinks
    // loads from the constant address space to their first use and the tile's set-up waits for the scalar cache (measured: +12 k cycles)
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+s"(w[i]));
    __builtin_memcpy(&dst, w, sizeof(T));
}

typedef __attribute__((ext_vector_type(4))) float mf32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t mu32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t mu32x2;

// One pointwise item: pixels [chunk * P, chunk * P + P) of one frame x the feature tiles of one channel block.  conv_pw.hip's scheme --
// weights of the block in LDS in fragment order, a wave owns 16 pixels at a time and loads their activations straight into MFMA B
// registers -- with the weights staged per ITEM (32-64 KB from L2 against >= 400 pixels x Cin of activations) and sc1 loads / stores.
// ACT: ACT_SILU compiled in, or -1 = read from a.act (one branch per TILE, never per element: see conv_halo_body.h finish_and_store).
template <typename E, int KS, int ACT>
__device__ __forceinline__ void ml_pw_tile(const MlPwDev& a, const int frame, const int chunk, const int cb, uint16_t* wl, const int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int nt0 = cb * a.NTL;
    const int ntl = a.NT - nt0 < a.NTL ? a.NT - nt0 : a.NTL;
    float* bl = reinterpret_cast<float*>(wl + (size_t)a.NTL * KS * 512);
    stage_lds16<256, 8>(wl, a.wfrag + (size_t)nt0 * KS * 512, ntl * KS * 64, tid);
    for (int i = tid; i < ntl * 16; i += 256) bl[i] = a.bias[nt0 * 16 + i];   // bias is padded to a multiple of 128 entries
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_up = __builtin_amdgcn_make_buffer_rsrc((void*)(a.up ? a.up : a.in), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0x7fffffff, 0x00020000);
    const int p_begin = chunk * a.P;
    const int p_end = p_begin + a.P < a.HW ? p_begin + a.P : a.HW;
    const int nmt = (p_end - p_begin + 15) >> 4;
    auto actf = [&](float v) {
        if constexpr (ACT >= 0) return h_act<ACT>(v);
        else return h_act_rt(a.act, v);
    };

    for (int mt = wave; mt < nmt; mt += 4) {
        const int pl = p_begin + mt * 16 + lrow;
        const bool ok = pl < p_end;
        const int pc = ok ? pl : p_begin;
        const uint32_t pix = (uint32_t)frame * (uint32_t)a.HW + (uint32_t)pc;
        const uint32_t ib = (pix * (uint32_t)a.in_cs + (uint32_t)(a.in_coff + kg * 8)) * 2u;
        uint32_t ub = 0;
        if (a.up_ks > 0) {   // workgroup-uniform
            const int oy = pc / a.W, ox = pc - oy * a.W;
            ub = (((uint32_t)frame * (uint32_t)a.up_HW + (uint32_t)((oy >> 1) * a.up_W + (ox >> 1))) * (uint32_t)a.up_cs + (uint32_t)(a.up_coff + kg * 8)) * 2u;
        }
        mu32x4 xb[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {   // no branch around a load: the (uniform) choice of the source tensor is a select of resource and offset
            const bool from_up = ks < a.up_ks;
            xb[ks] = __builtin_amdgcn_raw_buffer_load_b128(from_up ? r_up : r_in, (from_up ? ub : ib) + (uint32_t)ks * 64u, 0, 16);
        }
        const uint32_t ob = (pix * (uint32_t)a.out_cs + (uint32_t)(a.out_coff + nt0 * 16)) * 2u;
        int nt = 0;
        for (; nt + 1 < ntl; nt += 2) {
            mf32x4 acc0{0.f, 0.f, 0.f, 0.f}, acc1{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const mu32x4 w0 = *reinterpret_cast<const mu32x4*>(wl + ((size_t)(nt * KS + ks) * 64 + lane) * 8);
                const mu32x4 w1 = *reinterpret_cast<const mu32x4*>(wl + ((size_t)((nt + 1) * KS + ks) * 64 + lane) * 8);
                acc0 = E::mfma(w0, xb[ks], acc0);
                acc1 = E::mfma(w1, xb[ks], acc1);
            }
            const float4 b0 = *reinterpret_cast<const float4*>(bl + nt * 16 + kg * 4), b1 = *reinterpret_cast<const float4*>(bl + (nt + 1) * 16 + kg * 4);
            const uint32_t x0 = E::pack2(actf(acc0[0] + b0.x), actf(acc0[1] + b0.y)), x1 = E::pack2(actf(acc0[2] + b0.z), actf(acc0[3] + b0.w));
            const uint32_t y0 = E::pack2(actf(acc1[0] + b1.x), actf(acc1[1] + b1.y)), y1 = E::pack2(actf(acc1[2] + b1.z), actf(acc1[3] + b1.w));
            const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);   // conv_pw.hip / conv_halo's epilogue: a lane ends up with 8 consecutive channels
            const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
            const int c = (nt + (kg & 1)) * 16 + (kg >> 1) * 8;
            if (ok) {
                if (nt0 * 16 + c + 8 <= a.cout) __builtin_amdgcn_raw_buffer_store_b128(mu32x4{s0[0], s1[0], s0[1], s1[1]}, r_out, ob + (uint32_t)c * 2u, 0, 16);
                else if (nt0 * 16 + c + 4 <= a.cout) __builtin_amdgcn_raw_buffer_store_b64(mu32x2{s0[0], s1[0]}, r_out, ob + (uint32_t)c * 2u, 0, 16);
            }
        }
        for (; nt < ntl; ++nt) {   // the unpaired last feature tile: 8-byte stores
            mf32x4 acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const mu32x4 wf = *reinterpret_cast<const mu32x4*>(wl + ((size_t)(nt * KS + ks) * 64 + lane) * 8);
                acc = E::mfma(wf, xb[ks], acc);
            }
            const int c = nt * 16 + kg * 4;
            const float4 b4 = *reinterpret_cast<const float4*>(bl + c);
            const uint32_t q0 = E::pack2(actf(acc[0] + b4.x), actf(acc[1] + b4.y)), q1 = E::pack2(actf(acc[2] + b4.z), actf(acc[3] + b4.w));
            if (ok && nt0 * 16 + c < a.cout) __builtin_amdgcn_raw_buffer_store_b64(mu32x2{q0, q1}, r_out, ob + (uint32_t)c * 2u, 0, 16);
        }
    }
}

// Scratch instrumentation (-DADAS_ML_PROF, tools/ml_debug.py prof): shader-clock cycles of thread 0 per phase of the item loop, summed over
// the workgroup's items and added to ctl[2 + 2 * phase] (64-bit) when the workgroup leaves; ctl[14] counts items.
#ifdef ADAS_ML_PROF
#define MLPROF_INIT unsigned long long mlp_[6] = {0, 0, 0, 0, 0, 0}, mlp_t_ = __builtin_amdgcn_s_memtime(), mlp_n_ = 0;
#define MLPROF(i)                                                    \
    if (tid == 0) {                                                  \
        const unsigned long long t__ = __builtin_amdgcn_s_memtime(); \
        mlp_[i] += t__ - mlp_t_;                                     \
        mlp_t_ = t__;                                                \
    }
#define MLPROF_ITEM ++mlp_n_;
#define MLPROF_FLUSH                                                                                               \
    if (tid == 0) {                                                                                                \
        for (int i__ = 0; i__ < 6; ++i__) atomicAdd(reinterpret_cast<unsigned long long*>(g.ctl + 2) + i__, mlp_[i__]); \
        atomicAdd(reinterpret_cast<unsigned long long*>(g.ctl + 2) + 6, mlp_n_);                                   \
    }
#else
#define MLPROF_INIT
#define MLPROF(i)
#define MLPROF_ITEM
#define MLPROF_FLUSH
#endif

template <typename E, int KS>
__device__ __forceinline__ void ml_pw_dispatch(const MlPwDev& p, int frame, int chunk, int cb, uint16_t* lds, int tid) {
    if (p.act == ACT_SILU) ml_pw_tile<E, KS, ACT_SILU>(p, frame, chunk, cb, lds, tid);
    else ml_pw_tile<E, KS, -1>(p, frame, chunk, cb, lds, tid);
}

#ifdef ADAS_ML_OCC1   // experiment: one workgroup per CU, 512 registers per lane (spills go to AGPRs, not to scratch)
#define ADAS_ML_WAVES 1
#else
#define ADAS_ML_WAVES 2
#endif
template <typename E>
__global__ __launch_bounds__(256, ADAS_ML_WAVES) void conv_ml_kernel(MlArgs g) {
    E::enter();
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    // [0..15] the item's record, [16] its ticket -- BEHIND the tiles' scratch, not in front of it: the tile bodies' conflict-free LDS swizzles
    // assume the window starts at LDS offset 0 (a static __shared__ array in front shifts every ds_read_b128 by 80 bytes against the banks)
    uint32_t* s_rec = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(lds) + g.rec_off);
    const int tid = threadIdx.x;
    MLPROF_INIT
    // thread 0 carries the pipeline state: the ticket and the record of the item about to run
    unsigned mine = 0;
    mu32x4 rec[4] = {mu32x4{0, 0, 0, 0}, mu32x4{0, 0, 0, 0}, mu32x4{0, 0, 0, 0}, mu32x4{0, 0, 0, 0}};
    if (tid == 0) {
        mine = __hip_atomic_fetch_add(g.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mine < (unsigned)g.n_items) {
            const mu32x4* rp = reinterpret_cast<const mu32x4*>(g.items + mine);
#pragma unroll
            for (int i = 0; i < 4; ++i) rec[i] = rp[i];
        }
    }
    for (;;) {
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<mu32x4*>(s_rec + 4 * i) = rec[i];
            s_rec[16] = mine;
            s_rec[17] = 0;      // set by a waiter that gave up (its own time-out, or somebody else's error word): the item is skipped
        }
        __syncthreads();
        const unsigned t = __builtin_amdgcn_readfirstlane(s_rec[16]);
        MLPROF(0)   // ticket + record in hand (both fetched under the previous item) + broadcast
        if (t >= (unsigned)g.n_items) break;   // (uniform)
        const mu32x4 r0 = *reinterpret_cast<const mu32x4*>(s_rec);
        const int tile = (int)__builtin_amdgcn_readfirstlane(r0[0]);
        const unsigned where = __builtin_amdgcn_readfirstlane(r0[1]), kd = __builtin_amdgcn_readfirstlane(r0[2]);
        const unsigned arrive = __builtin_amdgcn_readfirstlane(r0[3]);
        const int layer = (int)(where & 255u), cb = (int)((where >> 8) & 255u), frame = (int)(where >> 16);
        const int kind = (int)(kd & 255u), n_dep = (int)(kd >> 8);
        unsigned next = 0;
        if (tid == 0) next = __hip_atomic_fetch_add(g.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the NEXT ticket: in flight under this item
        // the layer's descriptor through the scalar cache, in flight under the dependency wait
        MlLayerDev::U u;
        ml_copy_const(u, &g.layers[layer].u);
        MLPROF(1)
        // ---- wait until the frame is complete in every producer layer
        if (tid < n_dep) {
            const unsigned* c = g.ctl + s_rec[4 + tid];
            const unsigned target = s_rec[4 + ML_MAX_DEPS + tid];
            unsigned spins = 0;
            while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(8);
                ++spins;
                if (spins > (unsigned)g.spin_limit) {   // bounded: raise the error word (1 + ticket) and give the item up
                    __hip_atomic_store(g.ctl + 1, t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_rec[17] = 1;
                    break;
                }
                if ((spins & 255u) == 0u && __hip_atomic_load(g.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {   // somebody gave up: drain
                    s_rec[17] = 1;
                    break;
                }
            }
        }
        __syncthreads();
        // an item whose producers never completed computes nothing and publishes nothing (its consumers time out or drain in turn):
        // the launch ends with the error word set and adas_engine_ml_status / the pipeline's sync report it -- never a silently wrong tile
        const bool gave_up = __builtin_amdgcn_readfirstlane(s_rec[17]) != 0u;
        MLPROF(2)   // dependency wait
        // ---- the tile.  The bodies get an OPAQUE copy of the thread index: everything a tile derives from it (lane / wave decomposition, LDS swizzle
        // offsets, fragment addresses -- a dozen values per body, twelve bodies) is invariant across items, and LLVM hoists it all out of the
        // item loop, where it stays live across every body: the 252-register halo bodies then spill to scratch in their set-up (measured:
        // 15 k cycles of set-up per tile instead of 2.5 k).  Recomputing per item costs a few dozen VALU instructions.
        int tid_it = tid;
        asm volatile("" : "+v"(tid_it));
#ifdef ADAS_ML_ONEKIND   // code-size experiment: one halo body only
        if (!gave_up) halo_tile<E, 64, -1, 1, 256, true>(u.h, tile, cb, lds, tid_it);
#else
        if (gave_up) {
        } else if (kind == MLK_PW) {
            switch (u.p.KS) {
                case 2: ml_pw_dispatch<E, 2>(u.p, frame, tile, cb, lds, tid_it); break;
                case 3: ml_pw_dispatch<E, 3>(u.p, frame, tile, cb, lds, tid_it); break;
                case 4: ml_pw_dispatch<E, 4>(u.p, frame, tile, cb, lds, tid_it); break;
                case 6: ml_pw_dispatch<E, 6>(u.p, frame, tile, cb, lds, tid_it); break;
                case 8: ml_pw_dispatch<E, 8>(u.p, frame, tile, cb, lds, tid_it); break;
                case 12: ml_pw_dispatch<E, 12>(u.p, frame, tile, cb, lds, tid_it); break;
                default: ml_pw_dispatch<E, 16>(u.p, frame, tile, cb, lds, tid_it); break;
            }
        } else {
            switch (kind) {
                case MLK_H64_S1_256: halo_tile<E, 64, -1, 1, 256, true>(u.h, tile, cb, lds, tid_it); break;
                case MLK_H64_S1_128: halo_tile<E, 64, -1, 1, 128, true>(u.h, tile, cb, lds, tid_it); break;
                case MLK_H48_S1_256: halo_tile<E, 48, -1, 1, 256, true>(u.h, tile, cb, lds, tid_it); break;
                case MLK_H48_S1_128: halo_tile<E, 48, -1, 1, 128, true>(u.h, tile, cb, lds, tid_it); break;
                default: halo_tile<E, 64, -1, 2, 128, true>(u.h, tile, cb, lds, tid_it); break;
            }
        }
#endif
        MLPROF(3)   // tile body (thread 0's wave)
        // ---- the next item's record: its ticket came back under the tile; the 64 bytes arrive under the store drain
        if (tid == 0) {
            mine = next;
            if (mine < (unsigned)g.n_items) {
                const mu32x4* rp = reinterpret_cast<const mu32x4*>(g.items + mine);
#pragma unroll
                for (int i = 0; i < 4; ++i) rec[i] = rp[i];
            }
        }
        // ---- publish: the write-through stores of EVERY wave have left the CU, then one arrival
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        MLPROF(4)   // store drain + barrier
        if (tid == 0 && !gave_up) __hip_atomic_fetch_add(g.ctl + arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        MLPROF_ITEM
    }
    MLPROF_FLUSH
}

// ------------------------------------------------------------------------------------- grouped launch of independent layers
constexpr int GROUP_MAX = ML_GROUP_MAX;
struct GroupArgs {
    const MlLayerDev* layers;      // .kind + .u.h are read
    int n;
    int first_block[GROUP_MAX + 1];   // block range of layer l: [first_block[l], first_block[l + 1]), starts are multiples of 8 (XCD phase)
};

template <typename E>
__global__ __launch_bounds__(256, 2) void conv_halo_group_kernel(GroupArgs g) {
    E::enter();
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    int l = 0;
#pragma unroll
    for (int i = 1; i < GROUP_MAX; ++i)
        if (i < g.n && (int)blockIdx.x >= g.first_block[i]) l = i;
    const int local = (int)blockIdx.x - g.first_block[l];
    struct Head { int kind, n_dep, per_img, pad; } hd;
    ml_copy_const(hd, g.layers + l);
    HaloDev a;
    ml_copy_const(a, &g.layers[l].u.h);
    // conv_halo_kernel's workgroup -> (tile, channel block) map on the layer's own block range (its start is a multiple of 8: same XCD phase)
    const int xslot = local >> 3;
    const int xr = xslot / a.cbg;
    const int cb = (xr / a.tiles8) * a.cbg + (xslot - xr * a.cbg);
    const int tile = a.xmap ? (local & 7) * a.tiles8 + (xr % a.tiles8) : (xr % a.tiles8) * 8 + (local & 7);
    if (cb >= a.ncb || tile >= a.ntiles) return;
    switch (hd.kind) {
        case MLK_H64_S1_256: halo_tile<E, 64, -1, 1, 256, false>(a, tile, cb, lds, threadIdx.x); break;
        case MLK_H64_S1_128: halo_tile<E, 64, -1, 1, 128, false>(a, tile, cb, lds, threadIdx.x); break;
        case MLK_H48_S1_256: halo_tile<E, 48, -1, 1, 256, false>(a, tile, cb, lds, threadIdx.x); break;
        case MLK_H48_S1_128: halo_tile<E, 48, -1, 1, 128, false>(a, tile, cb, lds, threadIdx.x); break;
        case MLK_H32_S1_128: halo_tile<E, 32, -1, 1, 128, false>(a, tile, cb, lds, threadIdx.x); break;
        case MLK_H16_S1_128: halo_tile<E, 16, -1, 1, 128, false>(a, tile, cb, lds, threadIdx.x); break;
        default: halo_tile<E, 64, -1, 2, 128, false>(a, tile, cb, lds, threadIdx.x); break;
    }
}

#ifdef ADAS_HALO_PROF   // the halo tile's phase counters as accumulated by THIS translation unit's copy of g_halo_prof (tools/ml_hprof.py)
extern "C" int adas_debug_ml_halo_prof(unsigned long long* out16, int reset) {
    static unsigned long long h[256][16];
    if (out16) {
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_halo_prof), sizeof(h)) != hipSuccess) return -1;
        for (int i = 0; i < 16; ++i) {
            out16[i] = 0;
            for (int b = 0; b < 256; ++b) out16[i] += h[b][i];
        }
    }
    if (reset) {
        memset(h, 0, sizeof(h));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_halo_prof), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// ------------------------------------------------------------------------------------- host side
struct MlPlan {
    MlArgs args{};
    void* d_layers = nullptr;
    void* d_items = nullptr;
    void* d_ctl = nullptr;
    size_t ctl_bytes = 0, lds = 0;
    int grid = 0, prec = 0;
};

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

static int ml_halo_kind(int bn, int stride, int bm) {
    if (stride == 1 && bn == 64 && bm == 256) return MLK_H64_S1_256;
    if (stride == 1 && bn == 64 && bm == 128) return MLK_H64_S1_128;
    if (stride == 1 && bn == 48 && bm == 256) return MLK_H48_S1_256;
    if (stride == 1 && bn == 48 && bm == 128) return MLK_H48_S1_128;
    if (stride == 2 && bn == 64 && bm == 128) return MLK_H64_S2;
    return MLK_NONE;
}

// the grouped launch also carries the narrow blocks a batch-1 engine packs for (one frame at a time: every Detect conv is a 8-15 us
// latency chain of its own; twelve of them in two launches)
static int group_halo_kind(int bn, int stride, int bm) {
    if (stride == 1 && bn == 32 && bm == 128) return MLK_H32_S1_128;
    if (stride == 1 && bn == 16 && bm == 128) return MLK_H16_S1_128;
    return ml_halo_kind(bn, stride, bm);
}

static bool bytes_fit_31(const TView& v, int n) { return (double)n * v.h * v.w * v.cs * 2.0 < 2147483648.0; }

// feature tiles of one pointwise channel block: as many as fit 72 KB of LDS beside the bias (two workgroups per CU), an even count
static int ml_pw_ntl(int nt, int ks) {
    int ntl = 72 / ks;
    if (ntl >= nt) return nt;
    ntl &= ~1;
    return ntl < 2 ? 0 : ntl;
}

bool ml_layer_supported(const ConvArgs& a, int kernel) {
    if (!prec_is16(a.prec) || a.in.f32 || a.out.f32 || a.ds_w || a.n < 1 || a.n >= 65536) return false;
    if (a.act != ACT_NONE && a.act != ACT_SILU && a.act != ACT_RELU && a.act != ACT_LEAKY) return false;
    if (!bytes_fit_31(a.in, a.n) || !bytes_fit_31(a.out, a.n)) return false;
    if (a.res_mode != RES_NONE && (!a.res.p || a.res.f32 || !bytes_fit_31(a.res, a.n))) return false;
    if (kernel == CONV_HALO) {
        if (a.up_c > 0) return false;
        // launch_conv's order of choice: the persistent / stride-2 / LDS-DMA kernels come first and have no tile body here
        if (!(a.halo_bn > 0 && a.halo_bn != halo_bn(a.out.c))) {
            if (halo_rw_applicable(a.kh, a.kw, a.stride, a.pad, a.n, a.in, a.out)) return false;
            if (halo_s2p_applicable(a.kh, a.kw, a.stride, a.pad, a.res_mode, a.n, a.in, a.out)) return false;
            if (halo8_applicable(a.kh, a.kw, a.stride, a.pad, a.n, a.in, a.out, a.res, a.res_mode)) return false;
        }
        HaloDev d;
        int bn, bm;
        size_t lds;
        if (!halo_fill_dev(a, &d, &bn, &bm, &lds)) return false;
        return ml_halo_kind(bn, a.stride, bm) != MLK_NONE && lds <= 80 * 1024 - 256;   // two workgroups per CU with the record slot behind the scratch
    }
    if (kernel == CONV_PW) {
        if (a.kh != 1 || a.kw != 1 || a.stride != 1 || a.pad != 0 || a.res_mode != RES_NONE) return false;
        if ((a.in.c & 31) || ((a.in.cs | a.in.coff) & 7) || ((a.out.cs | a.out.coff) & 7) || (a.out.c & 3)) return false;
        const int ks = a.in.c / 32, nt = (a.out.c + 15) / 16;
        if (!(ks == 2 || ks == 3 || ks == 4 || ks == 6 || ks == 8 || ks == 12 || ks == 16)) return false;
        if (ml_pw_ntl(nt, ks) <= 0) return false;
        if (a.up_c > 0) {
            if ((a.up_c & 31) || a.up.c != a.up_c || 2 * a.up.h != a.in.h || 2 * a.up.w != a.in.w || a.up.f32 || ((a.up.cs | a.up.coff) & 7) || !bytes_fit_31(a.up, a.n)) return false;
        }
        return true;
    }
    return false;
}

namespace {
struct RView {
    const void* p;
    int c0, c1;
};
bool overlaps(const RView& a, const RView& b) { return a.p && a.p == b.p && a.c0 < b.c1 && b.c0 < a.c1; }

struct HostLayer {
    MlLayerDev dev;
    int kind = 0;
    int items_per_frame = 0;   // arrivals that complete one frame of this layer
    double item_us = 0.0;      // estimated duration of one item (list scheduler)
    std::vector<RView> reads;
    RView write{};
    std::vector<int> deps;     // after reduction
    uint64_t closure = 0;      // every layer whose frame is complete when this layer's frame is
    size_t lds = 0;
};
}  // namespace

MlPlan* ml_plan_create(const std::vector<ConvArgs>& layers, const std::vector<int>& kernels, int prec, std::string* why, MlPlanInfo* info, bool host_only) {
    auto fail = [&](const char* msg) -> MlPlan* {
        if (why) *why = msg;
        return nullptr;
    };
    const int NL = (int)layers.size();
    if (NL < 1 || NL > ML_MAX_LAYERS || kernels.size() != layers.size()) return fail("layer count");
    const int F = layers[0].n;
    std::vector<HostLayer> hl(NL);
    size_t lds_max = 0;
    for (int i = 0; i < NL; ++i) {
        const ConvArgs& a = layers[i];
        HostLayer& h = hl[i];
        if (a.n != F || a.prec != prec || !ml_layer_supported(a, kernels[i])) return fail("layer not supported");
        memset(&h.dev, 0, sizeof(h.dev));
        h.write = RView{a.out.p, a.out.coff, a.out.coff + a.out.c};
        if (kernels[i] == CONV_HALO) {
            int bn, bm;
            size_t lds;
            if (!halo_fill_dev(a, &h.dev.u.h, &bn, &bm, &lds)) return fail("halo plan");
            h.kind = ml_halo_kind(bn, a.stride, bm);
            h.dev.per_img = h.dev.u.h.NS * h.dev.u.h.TPS;
            h.items_per_frame = h.dev.per_img * h.dev.u.h.ncb;
            h.lds = lds;
            h.reads.push_back(RView{a.in.p, a.in.coff, a.in.coff + a.in.c});
            if (a.res_mode != RES_NONE) h.reads.push_back(RView{a.res.p, a.res.coff, a.res.coff + a.out.c});
            h.item_us = 2.5 + 2.0 * bm * bn * 9.0 * h.dev.u.h.cin_pad / 1.1e6;   // ~1.1 TFLOP/s per resident workgroup
        } else {
            MlPwDev& p = h.dev.u.p;
            h.kind = MLK_PW;
            p.in = (const uint16_t*)a.in.p; p.wfrag = (const uint16_t*)a.wgt; p.bias = a.bias; p.out = (uint16_t*)a.out.p;
            p.in_cs = a.in.cs; p.in_coff = a.in.coff; p.cin = a.in.c; p.out_cs = a.out.cs; p.out_coff = a.out.coff; p.cout = a.out.c;
            p.HW = a.out.h * a.out.w; p.W = a.out.w;
            const int nchunk = (p.HW + 511) / 512;
            p.P = ((p.HW + nchunk - 1) / nchunk + 15) / 16 * 16;
            p.chunks = (p.HW + p.P - 1) / p.P;
            p.KS = a.in.c / 32; p.NT = (a.out.c + 15) / 16; p.NTL = ml_pw_ntl(p.NT, p.KS); p.act = a.act;
            p.up = nullptr; p.up_cs = p.up_coff = p.up_ks = p.up_W = p.up_HW = 0;
            int c_lo = a.in.coff;
            if (a.up_c > 0) {
                p.up = (const uint16_t*)a.up.p; p.up_cs = a.up.cs; p.up_coff = a.up.coff; p.up_ks = a.up_c / 32; p.up_W = a.up.w; p.up_HW = a.up.h * a.up.w;
                h.reads.push_back(RView{a.up.p, a.up.coff, a.up.coff + a.up.c});
                c_lo += a.up_c;   // the leading channels come from the half-resolution tensor, not from the concat buffer
            }
            h.reads.push_back(RView{a.in.p, c_lo, a.in.coff + a.in.c});
            const int ncb = (p.NT + p.NTL - 1) / p.NTL;
            h.dev.per_img = p.chunks;
            h.items_per_frame = p.chunks * ncb;
            h.lds = (size_t)p.NTL * p.KS * 1024 + (size_t)p.NTL * 64;
            h.item_us = 3.0 + ((double)p.P * (a.in.c + p.NTL * 16) * 2.0 + (double)p.NTL * p.KS * 1024) / 12.0e3;   // ~12 GB/s per resident workgroup (6 TB/s over 512)
        }
        h.dev.kind = h.kind;
        lds_max = std::max(lds_max, h.lds);
    }
    if (lds_max > 80 * 1024 - 256) return fail("LDS");   // + the 128-byte record slot behind it, two workgroups per CU
    // ---- dependencies between the layers of the launch (frame granularity), reduced transitively
    for (int i = 0; i < NL; ++i) {
        uint64_t need = 0;
        for (int j = 0; j < i; ++j) {
            bool dep = false;
            for (auto& r : hl[i].reads) dep = dep || overlaps(r, hl[j].write);        // read after write
            for (auto& r : hl[j].reads) dep = dep || overlaps(r, hl[i].write);        // write after read (a buffer reused inside the launch)
            dep = dep || overlaps(hl[i].write, hl[j].write);                          // write after write
            if (dep) need |= 1ull << j;
        }
        uint64_t reduced = need;
        for (int j = 0; j < i; ++j)
            if (need & (1ull << j)) reduced &= ~(hl[j].closure & ~(1ull << j));   // whatever j's completion already implies need not be waited for
        hl[i].closure = 1ull << i;
        for (int j = 0; j < i; ++j)
            if (need & (1ull << j)) hl[i].closure |= hl[j].closure;
        for (int j = 0; j < i; ++j)
            if (reduced & (1ull << j)) hl[i].deps.push_back(j);
        if ((int)hl[i].deps.size() > ML_MAX_DEPS) return fail("too many producer layers for one item");
        hl[i].dev.n_dep = (int)hl[i].deps.size();
        for (int k = 0; k < hl[i].dev.n_dep; ++k) {
            hl[i].dev.dep_row[k] = hl[i].deps[k];
            hl[i].dev.dep_target[k] = hl[hl[i].deps[k]].items_per_frame;
        }
    }
    // ---- items, grouped by (layer, frame)
    struct Item { uint32_t tile; uint8_t layer, cb; uint16_t frame; };
    std::vector<std::vector<Item>> groups((size_t)NL * F);
    size_t n_items = 0;
    for (int i = 0; i < NL; ++i) {
        if (hl[i].kind == MLK_PW) {
            const MlPwDev& p = hl[i].dev.u.p;
            const int ncb = (p.NT + p.NTL - 1) / p.NTL;
            for (int f = 0; f < F; ++f)
                for (int ch = 0; ch < p.chunks; ++ch)
                    for (int cb = 0; cb < ncb; ++cb) groups[(size_t)i * F + f].push_back(Item{(uint32_t)ch, (uint8_t)i, (uint8_t)cb, (uint16_t)f});
        } else {
            const HaloDev& d = hl[i].dev.u.h;
            if (d.ncb > 255) return fail("channel blocks");
            for (int f = 0; f < F; ++f)
                for (int t = 0; t < hl[i].dev.per_img; ++t)
                    for (int cb = 0; cb < d.ncb; ++cb) groups[(size_t)i * F + f].push_back(Item{(uint32_t)(f * hl[i].dev.per_img + t), (uint8_t)i, (uint8_t)cb, (uint16_t)f});
        }
        n_items += (size_t)hl[i].items_per_frame * F;
    }
    if (n_items >= (1u << 30)) return fail("item count");
    // resident workgroups: two per CU (256 CUs on MI355X); host-only planning (no device) assumes that chip
    int cus = 256;
    if (!host_only) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        else (void)hipGetLastError();
    }
    const int grid = std::max(1, std::min(env_int("ADAS_ML_GRID", 2 * cus), (int)n_items));
    // ---- order.  0: layer-major (launch order of the layers, frames ascending).  1 (default): list schedule -- simulate `grid` workgroups
    // taking the ready (layer, frame) group with the longest remaining path first; a group is ready when its producers' groups of the same
    // frame have FINISHED in the simulation, so every item's producers hold smaller tickets by construction (checked below anyway).
    const int order_mode = env_int("ADAS_ML_ORDER", 1);
    std::vector<Item> order;
    order.reserve(n_items);
    if (order_mode == 0) {
        for (int i = 0; i < NL; ++i)
            for (int f = 0; f < F; ++f)
                for (auto& it : groups[(size_t)i * F + f]) order.push_back(it);
    } else {
        std::vector<double> bottom(NL, 0.0);   // longest path (estimated us) from a layer's item to the end of the launch
        for (int i = NL - 1; i >= 0; --i) {
            double b = 0.0;
            for (int j = i + 1; j < NL; ++j)
                if (std::find(hl[j].deps.begin(), hl[j].deps.end(), i) != hl[j].deps.end()) b = std::max(b, bottom[j]);
            bottom[i] = b + hl[i].item_us;
        }
        std::vector<int> pending((size_t)NL * F), remaining((size_t)NL * F), next_item((size_t)NL * F, 0);
        struct Ready { double prio; int frame, layer; };
        auto worse = [](const Ready& a, const Ready& b) {
            if (a.prio != b.prio) return a.prio < b.prio;
            if (a.frame != b.frame) return a.frame > b.frame;
            return a.layer > b.layer;
        };
        std::priority_queue<Ready, std::vector<Ready>, decltype(worse)> ready(worse);
        for (int i = 0; i < NL; ++i)
            for (int f = 0; f < F; ++f) {
                pending[(size_t)i * F + f] = (int)hl[i].deps.size();
                remaining[(size_t)i * F + f] = (int)groups[(size_t)i * F + f].size();
                if (hl[i].deps.empty()) ready.push(Ready{bottom[i], f, i});
            }
        std::vector<std::vector<int>> consumers(NL);
        for (int i = 0; i < NL; ++i)
            for (int d : hl[i].deps) consumers[d].push_back(i);
        struct Running { double end; int layer, frame; };
        auto later = [](const Running& a, const Running& b) { return a.end > b.end; };
        std::priority_queue<Running, std::vector<Running>, decltype(later)> running(later);
        double now = 0.0;
        int free_slots = grid;
        while (order.size() < n_items) {
            while (free_slots > 0 && !ready.empty()) {
                const Ready r = ready.top();
                const size_t gi = (size_t)r.layer * F + r.frame;
                order.push_back(groups[gi][next_item[gi]++]);
                running.push(Running{now + hl[r.layer].item_us, r.layer, r.frame});
                --free_slots;
                if (next_item[gi] == (int)groups[gi].size()) ready.pop();
            }
            if (running.empty()) return fail("scheduler stalled (cyclic dependencies?)");
            const Running done = running.top();
            running.pop();
            now = done.end;
            ++free_slots;
            const size_t gi = (size_t)done.layer * F + done.frame;
            if (--remaining[gi] == 0)
                for (int c : consumers[done.layer])
                    if (--pending[(size_t)c * F + done.frame] == 0) ready.push(Ready{bottom[c], done.frame, c});
        }
    }
    // ---- the invariant the kernel's progress argument rests on: every item comes after ALL items of its producers' groups
    {
        std::vector<int> seen((size_t)NL * F, 0);
        for (auto& it : order) {
            for (int d : hl[it.layer].deps)
                if (seen[(size_t)d * F + it.frame] != hl[d].items_per_frame) return fail("item order violates a dependency");
            ++seen[(size_t)it.layer * F + it.frame];
        }
        for (int i = 0; i < NL; ++i)
            for (int f = 0; f < F; ++f)
                if (seen[(size_t)i * F + f] != hl[i].items_per_frame) return fail("item table incomplete");
    }
    std::vector<MlItemRec> words(n_items);
    for (size_t k = 0; k < n_items; ++k) {
        const Item& it = order[k];
        const HostLayer& h = hl[it.layer];
        MlItemRec r;
        memset(&r, 0, sizeof(r));
        r.tile = it.tile;
        r.where = (uint32_t)it.layer | ((uint32_t)it.cb << 8) | ((uint32_t)it.frame << 16);
        r.kind = (uint32_t)h.kind | ((uint32_t)h.deps.size() << 8);
        r.arrive = (uint32_t)(ML_CTL_HEAD + (size_t)it.layer * F + it.frame);
        for (size_t d = 0; d < h.deps.size(); ++d) {
            r.dep_idx[d] = (uint32_t)(ML_CTL_HEAD + (size_t)h.deps[d] * F + it.frame);
            r.dep_target[d] = (uint32_t)hl[h.deps[d]].items_per_frame;
        }
        words[k] = r;
    }
    if (info) {
        info->n_layers = NL; info->n_items = (int)n_items; info->frames = F; info->grid = grid; info->order = order_mode; info->lds = lds_max;
        info->items_per_layer.clear(); info->deps.clear(); info->targets.clear(); info->item_words.clear();
        for (int i = 0; i < NL; ++i) {
            info->items_per_layer.push_back(hl[i].items_per_frame * F);
            info->deps.push_back(hl[i].deps);
            std::vector<int> tg;
            for (int d : hl[i].deps) tg.push_back(hl[d].items_per_frame);
            info->targets.push_back(tg);
        }
        for (auto& w : words) info->item_words.push_back((uint64_t)w.tile | ((uint64_t)w.where << 32));
    }
    MlPlan* pl = new MlPlan();
    pl->prec = prec;
    pl->grid = grid;
    pl->args.rec_off = (int)((lds_max + 127) / 128 * 128);
    pl->lds = (size_t)pl->args.rec_off + 128;
    pl->ctl_bytes = ((size_t)ML_CTL_HEAD + (size_t)NL * F) * 4;
    pl->args.n_items = (int)n_items;
    pl->args.frames = F;
    pl->args.spin_limit = env_int("ADAS_ML_SPIN", 1 << 19);
    if (host_only) return pl;
    std::vector<MlLayerDev> devs(NL);
    for (int i = 0; i < NL; ++i) devs[i] = hl[i].dev;
    bool ok = hipMalloc(&pl->d_layers, devs.size() * sizeof(MlLayerDev)) == hipSuccess && hipMalloc(&pl->d_items, words.size() * sizeof(MlItemRec)) == hipSuccess &&
              hipMalloc(&pl->d_ctl, pl->ctl_bytes) == hipSuccess;
    ok = ok && hipMemcpy(pl->d_layers, devs.data(), devs.size() * sizeof(MlLayerDev), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(pl->d_items, words.data(), words.size() * sizeof(MlItemRec), hipMemcpyHostToDevice) == hipSuccess && hipMemset(pl->d_ctl, 0, pl->ctl_bytes) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        ml_plan_destroy(pl);
        return fail("device allocation");
    }
    pl->args.layers = (const MlLayerDev*)pl->d_layers;
    pl->args.items = (const MlItemRec*)pl->d_items;
    pl->args.ctl = (unsigned*)pl->d_ctl;
    return pl;
}

void ml_plan_destroy(MlPlan* p) {
    if (!p) return;
    if (p->d_layers) (void)hipFree(p->d_layers);
    if (p->d_items) (void)hipFree(p->d_items);
    if (p->d_ctl) (void)hipFree(p->d_ctl);
    delete p;
}

hipError_t ml_launch(const MlPlan* p, hipStream_t st) {
    if (!p || !p->d_ctl) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_ml_kernel<Fp16>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_ml_kernel<Bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_done = true;
    }
    hipError_t e = hipMemsetAsync(p->d_ctl, 0, p->ctl_bytes, st);   // ticket, error word and every arrival counter: zero before EVERY launch (a memset node under capture)
    if (e != hipSuccess) return e;
    if (p->prec == PREC_FP16) hipLaunchKernelGGL(conv_ml_kernel<Fp16>, dim3(p->grid), dim3(256), p->lds, st, p->args);
    else hipLaunchKernelGGL(conv_ml_kernel<Bf16>, dim3(p->grid), dim3(256), p->lds, st, p->args);
    return hipGetLastError();
}

int ml_plan_status(const MlPlan* p, unsigned* error_word, unsigned* head16) {
    if (!p || !p->d_ctl || !error_word) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    unsigned w[ML_CTL_HEAD];
    if (hipMemcpy(w, p->d_ctl, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    *error_word = w[1];
    if (head16) memcpy(head16, w, sizeof(w));
    return 0;
}

// ------------------------------------------------------------------------------------- grouped launch: host side
bool group_layer_supported(const ConvArgs& a, int kernel) {
    if (kernel != CONV_HALO || !prec_is16(a.prec) || a.in.f32 || a.out.f32 || a.ds_w || a.up_c > 0) return false;
    if (a.act != ACT_NONE && a.act != ACT_SILU && a.act != ACT_RELU && a.act != ACT_LEAKY) return false;
    if (!(a.halo_bn > 0 && a.halo_bn != halo_bn(a.out.c))) {   // launch_conv's order of choice
        if (halo_rw_applicable(a.kh, a.kw, a.stride, a.pad, a.n, a.in, a.out)) return false;
        if (halo_s2p_applicable(a.kh, a.kw, a.stride, a.pad, a.res_mode, a.n, a.in, a.out)) return false;
        if (halo8_applicable(a.kh, a.kw, a.stride, a.pad, a.n, a.in, a.out, a.res, a.res_mode)) return false;
    }
    HaloDev d;
    int bn, bm;
    size_t lds;
    if (!halo_fill_dev(a, &d, &bn, &bm, &lds)) return false;
    return group_halo_kind(bn, a.stride, bm) != MLK_NONE && lds <= 80 * 1024;
}

std::vector<int> ml_levels(const std::vector<ConvArgs>& layers) {
    const int n = (int)layers.size();
    std::vector<int> level(n, 0);
    auto reads = [&](const ConvArgs& a) {
        std::vector<RView> r;
        int c_lo = a.in.coff;
        if (a.up_c > 0) { r.push_back(RView{a.up.p, a.up.coff, a.up.coff + a.up.c}); c_lo += a.up_c; }
        r.push_back(RView{a.in.p, c_lo, a.in.coff + a.in.c});
        if (a.res_mode != RES_NONE) r.push_back(RView{a.res.p, a.res.coff, a.res.coff + a.out.c});
        return r;
    };
    for (int i = 0; i < n; ++i) {
        const RView wi{layers[i].out.p, layers[i].out.coff, layers[i].out.coff + layers[i].out.c};
        const auto ri = reads(layers[i]);
        for (int j = 0; j < i; ++j) {
            const RView wj{layers[j].out.p, layers[j].out.coff, layers[j].out.coff + layers[j].out.c};
            bool dep = overlaps(wi, wj);
            for (auto& r : ri) dep = dep || overlaps(r, wj);
            for (auto& r : reads(layers[j])) dep = dep || overlaps(r, wi);
            if (dep && level[j] + 1 > level[i]) level[i] = level[j] + 1;
        }
    }
    return level;
}

struct MlGroup {
    GroupArgs args{};
    void* d_layers = nullptr;
    size_t lds = 0;
    int grid = 0, prec = 0;
};

MlGroup* ml_group_create(const std::vector<ConvArgs>& layers, int prec, std::string* why) {
    auto fail = [&](const char* msg) -> MlGroup* {
        if (why) *why = msg;
        return nullptr;
    };
    const int n = (int)layers.size();
    if (n < 2 || n > GROUP_MAX) return fail("layer count");
    {
        const std::vector<int> lv = ml_levels(layers);
        for (int v : lv)
            if (v != 0) return fail("layers of a grouped launch must be independent");
    }
    std::vector<MlLayerDev> devs(n);
    MlGroup* g = new MlGroup();
    g->prec = prec;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const ConvArgs& a = layers[i];
        memset(&devs[i], 0, sizeof(MlLayerDev));
        int bn, bm;
        size_t lds;
        if (a.prec != prec || !group_layer_supported(a, CONV_HALO) || !halo_fill_dev(a, &devs[i].u.h, &bn, &bm, &lds)) {
            delete g;
            return fail("layer not supported");
        }
        devs[i].kind = group_halo_kind(bn, a.stride, bm);
        const HaloDev& d = devs[i].u.h;
        g->args.first_block[i] = blocks;
        blocks += 8 * d.tiles8 * d.cbg * ((d.ncb + d.cbg - 1) / d.cbg);   // conv_halo's grid for this layer: a multiple of 8
        g->lds = std::max(g->lds, lds);
    }
    for (int i = n; i <= GROUP_MAX; ++i) g->args.first_block[i] = blocks;
    g->args.n = n;
    g->grid = blocks;
    if (hipMalloc(&g->d_layers, devs.size() * sizeof(MlLayerDev)) != hipSuccess ||
        hipMemcpy(g->d_layers, devs.data(), devs.size() * sizeof(MlLayerDev), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        ml_group_destroy(g);
        return fail("device allocation");
    }
    g->args.layers = (const MlLayerDev*)g->d_layers;
    return g;
}

void ml_group_destroy(MlGroup* g) {
    if (!g) return;
    if (g->d_layers) (void)hipFree(g->d_layers);
    delete g;
}

hipError_t ml_group_launch(const MlGroup* g, hipStream_t st) {
    if (!g || !g->d_layers) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_halo_group_kernel<Fp16>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_halo_group_kernel<Bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_done = true;
    }
    if (g->prec == PREC_FP16) hipLaunchKernelGGL(conv_halo_group_kernel<Fp16>, dim3(g->grid), dim3(256), g->lds, st, g->args);
    else hipLaunchKernelGGL(conv_halo_group_kernel<Bf16>, dim3(g->grid), dim3(256), g->lds, st, g->args);
    return hipGetLastError();
}

}  // namespace adas
