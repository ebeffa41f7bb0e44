// conv_halo8.hip -- stride-1 3x3 convolution for Cout % 128 == 0: a persistent, LDS-DMA fed, two-wave-group form of the halo kernel.
//
// conv_halo.hip stages every 32-channel chunk through registers and synchronises twice per chunk with nothing in flight across
// the barriers; it tops out at ~0.9-1.0 PFLOP/s.  This kernel keeps the tiling (strip-linear
// 256-pixel tiles, 64-byte swizzled window pixels, swizzled 64-byte weight rows, the same CONV_HALO weight packing) and changes
// the synchronisation structure:
//   * one 8-wave workgroup per CU owns 256 pixels x 128 output channels (waves 0-3 / 4-7 = the two 64-channel halves, each wave
//     4 x 4 MFMA tiles) and walks a list of (tile, channel block) items: the stream of taps never stops at a tile boundary;
//   * window and weights arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave instruction): no staging registers, no
//     ds_write pass.  The swizzle is applied on the SOURCE address (lane l of a piece fetches the 16 bytes that belong at LDS
//     position l), out-of-image window pixels are out-of-range buffer offsets (the DMA writes zeros);
//   * LDS holds two window buffers (chunk c is read while chunk c+1 lands) and a 9-slot ring of per-tap weight tiles (2 x 64 rows
//     x 64 B); at tap T every wave issues its 1 KiB of the weights of tap T+4 and, on taps 1-5 of a chunk, 1 KiB of the next
//     chunk's (or the next item's first) window.  Waits are COUNTED (`s_waitcnt vmcnt(N)`, N = the pieces issued in the last
//     three taps): nothing ever drains the queue inside the stream;
//   * MODE 1: the two wave groups run one barrier apart (group 1 takes one extra barrier up front): while one group issues the 8
//     fragment reads + DMA pieces of a tap, the other group's 16 MFMAs of the previous tap occupy the matrix pipe of the same SIMDs;
//     MODE 2: one barrier per tap row, weights six taps ahead, all waves free-running in between (layers with few chunks per item);
//   * an item's epilogue costs 2-4 k cycles with nothing to hide it under, so everything else that belongs to an item boundary is
//     moved off it: the residual and the next item's bias are fetched under the last tap row (counted waits widened by their 12
//     queue slots), the next item's tap / window offsets are computed between the last chunk's MFMAs.
// Ordering rules the schedule relies on (cdna_hip_programming.md, "256^2 8-phase template"): a DMA piece is visible to a reader
// that has passed a barrier which every issuing wave reached after its counted wait; with the groups one barrier apart that is
// "wait at the end of tap T's read segment, read in tap T+1's".  A buffer is re-filled no earlier than the read segment two taps
// after its last read.
#include "kernels.h"
#include "elem16.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float qf32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t qu32x4;
typedef __attribute__((address_space(3))) void* lds_vp;

struct H8Dev {
    const uint16_t* in;
    const uint16_t* wgt;
    const float* bias;
    uint16_t* out;
    const uint16_t* res;
    uint32_t in_bytes, wgt_bytes, out_bytes, res_bytes;
    int in_cs, in_coff, cin, H, W;
    int out_cs, out_coff, cout;
    int res_cs, res_coff, res_mode;
    int nchunk;               // 32-channel chunks of the packed weights
    int SW, NS, TPS, WW;      // strip width, strips per row, tiles per strip, window width (conv_halo.hip's plan)
    uint32_t mg_ww, mg_sw;
    uint32_t mg_img, mg_tps, mg_upt;  // ceil(2^32 / d) for d = NS * TPS, TPS, ncb / cpw: unit -> tile -> (image, strip) by multiply-high
    int ntiles, tiles8, ncb, cpw;  // tiles, ceil(tiles / 8) (one contiguous range per XCD), 128-channel blocks
    // projection shortcut folded in (ResNet layerN.0: out = relu(conv3x3(t) + W_ds x[2y, 2x] + biases)): the 1x1 stride-2 conv is
    // ndc extra 32-channel K steps accumulated before the 3x3 stream starts (ndc = 0: none)
    const uint16_t* ds_in;    // x: [n][2H][2W][ds_cs]
    const uint16_t* ds_w;     // [cout / 64][ndc][64 rows][32]
    const float* ds_bias;
    uint32_t ds_bytes;
    int ds_cs, ds_coff, ds_H, ds_W, ndc;
};

constexpr int H8_THR = 512;
constexpr int H8_BM = 256;
constexpr int H8_MAXPIX = 640;
static int h8_policy() {   // ADAS_H8_PLAN: strip-width policy of plan_halo (kernels.h): 1 = smallest window among the most efficient widths
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_H8_PLAN");
        v = e ? atoi(e) : 0;
    }
    return v;
}
constexpr int H8_WIN = H8_MAXPIX * 64;          // bytes of one window buffer
constexpr int H8_TAP = 2 * 64 * 64;             // bytes of one tap's weights (two 64-row blocks)
constexpr int H8_WR = 2 * H8_WIN;               // byte offset of the weight ring
constexpr int H8_LDS = H8_WR + 9 * H8_TAP;      // 155,648 B
constexpr int H8_NWP = H8_MAXPIX / 16 / 8;      // window pieces per wave (5)
constexpr int H8_SLAB = 9 * 64 * 64;            // bytes of one (64-channel block, chunk) weight slab
constexpr uint32_t H8_OOB = 0xF0000000u;
constexpr int H8_SLOTS = 32;                    // workgroups per XCD = CUs per XCD (MI355X: 8 x 32)

// Synchronisation variants (template MODE): 1 = two barriers per tap, the two wave groups one barrier apart (the read segment of
// one group runs under the MFMA segment of the other); 2 = one barrier per tap ROW (three taps), all waves free-running in
// between.  Weight pieces are issued `look` taps ahead of their first read.  Measured in the lane network at 64 frames
// (profiles/r02/h8_modes.txt): 1 wins from 16 chunks per item up (512 -> 512: 0.369 vs 0.389 ms for the three layers), 2 below
// (128 -> 128: 0.494 vs 0.551 ms); a one-barrier-per-tap free-running form was never the best and is not kept.
__host__ __device__ constexpr int h8_look(int mode) { return mode == 2 ? 6 : 4; }

template <int N>
__device__ __forceinline__ void h8_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int ACT>
__device__ __forceinline__ float h8_act(float v) {
    if (ACT == ACT_SILU) return v * fast_rcp(1.0f + __expf(-v));
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_LEAKY) return fmaxf(v, 0.1f * v);
    return v;
}

// pieces a wave issues in the read segment of tap k of a chunk: one weight piece, plus one window piece on taps 1..5
__host__ __device__ constexpr int h8_issued(int k) { return 1 + ((k >= 1 && k <= H8_NWP) ? 1 : 0); }
// pieces that may still be in flight after the wait of tap k: per-tap barriers -> those issued in taps k-2, k-1, k (the piece a
// tap k+1 read needs was issued at k+1-4); per-row barriers -> those issued in k's own row (row R+1 reads what row R-1 issued)
__host__ __device__ constexpr int h8_allow(int mode, int k) {
    return mode == 2 ? h8_issued(k) + h8_issued(k - 1) + h8_issued(k - 2)
                     : h8_issued(k) + h8_issued((k + 8) % 9) + h8_issued((k + 7) % 9);
}

#ifdef ADAS_H8_PROF   // scratch instrumentation (tools/experiments/h8_prof.py): shader cycles of waves 0 and 4 per item phase
__device__ unsigned long long g_h8_prof[256][32];
#define H8P(i)                                      \
    if (lane == 0 && grp == 0) {                    \
        const unsigned long long t__ = clock64();   \
        pacc__[i] += t__ - tprev__;                 \
        tprev__ = t__;                              \
    }
#else
#define H8P(i)
#endif

template <typename E, int ACT, int MODE>
__global__ __launch_bounds__(H8_THR, 1) void conv_h8_kernel(H8Dev a) {
    E::enter();
    typedef typename E::vec8 hvec8;
    constexpr int LOOK = h8_look(MODE);
    extern __shared__ __attribute__((aligned(16))) uint8_t lds8[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, kg = lane >> 4;
    const int hb = wave >> 2, grp = wave & 3;   // 64-channel half (= wave group), pixel quarter
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    // Work list: XCD x owns the contiguous tile range [x * tiles8, ...).  A unit = one tile x `cpw` consecutive 128-channel blocks
    // (the host picks cpw, a divisor of ncb, so that the units fill the chip); workgroup `slot` takes units slot, slot + nslot, ...
    // and computes a unit's channel blocks back to back (an "item" = one tile x one block): the tile's window is re-read by the
    // CU that just read it, its addresses are computed once per unit.  Units of one tile sit in neighbouring slots.
    int tiles_here = a.ntiles - xcd * a.tiles8;
    tiles_here = tiles_here < 0 ? 0 : (tiles_here > a.tiles8 ? a.tiles8 : tiles_here);
    const int upt = a.ncb / a.cpw;            // units per tile
    const int units_here = tiles_here * upt;
    if (slot >= units_here) return;

    __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rwg = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, a.wgt_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.out_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res_mode != RES_NONE ? a.res : a.out), 0,
                                                                    a.res_mode != RES_NONE ? a.res_bytes : 0u, 0x00020000);
    const int per_img = a.NS * a.TPS;
    const int gsw[4] = {0, 2, 3, 1};
    // lane part of a weight piece: rows (lane >> 2) of the wave's 16-row group, 16-byte position swizzled by the row key
    const uint32_t wlane = (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ gsw[(lane >> 4) & 3]) << 4));
    const uint32_t wpiece = (uint32_t)((lane & 3) ^ (((lane >> 4) & 1) << 1)) << 4;   // window: position -> source 16-byte chunk
    // per-lane fragment read offset into the weight ring (bytes)
    const uint32_t wrd = (uint32_t)(H8_WR + hb * 4096 + lrow * 64 + ((kg ^ gsw[(lrow >> 2) & 3]) << 4));
    // epilogue: after the 16-lane row exchange a lane owns 8 consecutive channels: tile i + (kg & 1), channels (kg >> 1) * 8 ..
    const uint32_t ch_lane = (uint32_t)((hb * 64 + (kg & 1) * 16 + (kg >> 1) * 8) * 2);
    const bool has_res = a.res_mode != RES_NONE;

    struct Tile {
        int img, sx0, p0, y_first;
    };
    auto decode = [&](int u) {   // u: unit index within the XCD's range.  n / d = umulhi(n, ceil(2^32 / d)) for n * d < 2^32
        Tile t;
        int tile = xcd * a.tiles8 + (upt == 1 ? u : (int)__umulhi((uint32_t)u, a.mg_upt));
        t.img = per_img == 1 ? tile : (int)__umulhi((uint32_t)tile, a.mg_img);
        tile -= t.img * per_img;
        const int strip = a.TPS == 1 ? tile : (int)__umulhi((uint32_t)tile, a.mg_tps);
        t.sx0 = strip * a.SW;
        t.p0 = (tile - strip * a.TPS) * H8_BM;
        t.y_first = (int)(((uint32_t)t.p0 * a.mg_sw) >> 20);
        return t;
    };
    // source byte offset of this lane's 16 bytes in the wave's window piece i (chunk 0); H8_OOB -> the DMA writes zeros
    auto win_offset = [&](const Tile& t, int i) {
        const int y_lastp = (int)(((uint32_t)(t.p0 + H8_BM - 1) * a.mg_sw) >> 20);
        const int npix = (y_lastp - t.y_first + 3) * a.WW;
        const int pix = (wave + 8 * i) * 16 + (lane >> 2);
        const int wy = (int)(((uint32_t)pix * a.mg_ww) >> 20), wx = pix - wy * a.WW;
        const int iy = t.y_first - 1 + wy, ix = t.sx0 - 1 + wx;
        const bool ok = pix < npix && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const uint32_t off = ((uint32_t)((t.img * a.H + iy) * a.W + ix) * (uint32_t)a.in_cs + (uint32_t)a.in_coff) * 2u + wpiece;
        const uint32_t m = 0u - (uint32_t)ok;   // select by mask: a branch here would split the MFMA block this is scheduled into
        return (off & m) | (H8_OOB & ~m);
    };
    // (window pixel of this lane's output pixel j at tap (0, 0)) * 64 + kg * 16, and from it the byte offset within a window
    // buffer of the lane's 16 bytes at tap t: pixel pw = that + tap shift, 16-byte position kg ^ (((pw >> 2) & 1) << 1)
    auto tap00 = [&](const Tile& t, int j) {
        const int p = t.p0 + (grp * 4 + j) * 16 + lrow;
        const int y = (int)(((uint32_t)p * a.mg_sw) >> 20), xs = p - y * a.SW;
        return (uint32_t)((((y - t.y_first) * a.WW + xs) << 6) | (kg << 4));
    };
    auto tap_offset = [&](uint32_t ap64, int t) {
        const uint32_t v = ap64 + (uint32_t)(((t / 3) * a.WW + (t % 3)) << 6);
        return v ^ ((v >> 3) & 0x20u);
    };
    // pixel index of the lane's output pixel j (H8_OOB outside the image: loads return 0, stores are dropped)
    auto out_pixel = [&](const Tile& t, int j) {
        const int p = t.p0 + (grp * 4 + j) * 16 + lrow;
        const int y = (int)(((uint32_t)p * a.mg_sw) >> 20), x = t.sx0 + (p - y * a.SW);
        return (y < a.H && x < a.W) ? (uint32_t)((t.img * a.H + y) * a.W + x) : H8_OOB;
    };
    // scalar byte offset of this wave's 16 rows of (channel block cb, half hb, chunk 0, tap 0)
    auto wgt_base = [&](int cb) { return (uint32_t)(((2 * cb + hb) * a.nchunk) * H8_SLAB + grp * 1024); };
    auto load_bias = [&](int cb, float4* b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b[i] = *reinterpret_cast<const float4*>(a.bias + cb * 128 + hb * 64 + i * 16 + kg * 4);
            if (a.ndc > 0) {
                const float4 d = *reinterpret_cast<const float4*>(a.ds_bias + cb * 128 + hb * 64 + i * 16 + kg * 4);
                b[i].x += d.x; b[i].y += d.y; b[i].z += d.z; b[i].w += d.w;
            }
        }
    };

    auto first_cb = [&](int u) { return upt == 1 ? 0 : (u - (int)__umulhi((uint32_t)u, a.mg_upt) * upt) * a.cpw; };
    int ti = slot, cb = first_cb(slot), cbi = 0;   // unit, channel block, its index within the unit
    Tile cur = decode(ti);
    uint32_t gcur[H8_NWP], gnxt[H8_NWP];   // window piece sources: the chunk being fetched / chunk 0 of the next item
    uint32_t xoff[4][9];                   // tap offsets of the item being computed; re-written tap by tap during its last chunk
    uint32_t po[4];
#pragma unroll
    for (int i = 0; i < H8_NWP; ++i) gcur[i] = win_offset(cur, i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t ap64 = tap00(cur, j);
#pragma unroll
        for (int t = 0; t < 9; ++t) xoff[j][t] = tap_offset(ap64, t);
        po[j] = out_pixel(cur, j);
    }
    uint32_t wcur = wgt_base(cb);
    int par = 0;
    float4 biasn[4];   // bias of the item about to start
    load_bias(cb, biasn);

    // ---- prologue: window of chunk 0 into buffer 0, weights of taps 0 .. LOOK-1
#pragma unroll
    for (int i = 0; i < H8_NWP; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_vp)(lds8 + (wave + 8 * i) * 1024), 16, gcur[i], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < LOOK; ++t)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwg, (lds_vp)(lds8 + H8_WR + t * H8_TAP + wave * 1024), 16, wlane, wcur + t * 4096, 0, 0);
    qf32x4 acc[4][4];   // starts at the bias
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = qf32x4{biasn[i].x, biasn[i].y, biasn[i].z, biasn[i].w};
    h8_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (MODE == 1 && hb) __builtin_amdgcn_s_barrier();   // group 1 runs one segment behind group 0

#ifdef ADAS_H8_PROF
    unsigned long long tprev__ = clock64();
    unsigned long long pacc__[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int nit__ = 0;
#endif
    for (;;) {
        H8P(5)
        const bool newtile = cbi + 1 == a.cpw;   // the item after this one starts another unit
        const bool has_next = !newtile || ti + nslot < units_here;
        const int cbn = newtile ? first_cb(has_next ? ti + nslot : ti) : cb + 1;
        const uint32_t ch0 = (uint32_t)(cb * 256) + ch_lane;
        Tile nxt = cur;   // decoded at the start of the last chunk, with everything else the next item needs
        uint32_t wnxt_item = 0, apn[4] = {0, 0, 0, 0};
        qu32x4 rraw[4][2];   // residual, fetched under the last row of taps of the last chunk
        H8P(0)

        if (a.ndc > 0) {
            // ---- projection shortcut: ndc K steps of x[2y, 2x] (no halo: the tile's own 256 pixels, 16 KB per step) against
            // [128 x 32] weight tiles, staged in the window buffer the first chunk does not use, both wave groups in lockstep
            // (group 0 waits one barrier for group 1, the stagger is re-established afterwards)
            if (MODE == 1 && !hb) __builtin_amdgcn_s_barrier();
            __amdgpu_buffer_rsrc_t rds = __builtin_amdgcn_make_buffer_rsrc((void*)a.ds_in, 0, a.ds_bytes, 0x00020000);
            __amdgpu_buffer_rsrc_t rdw = __builtin_amdgcn_make_buffer_rsrc((void*)a.ds_w, 0, (uint32_t)(a.cout / 64) * (uint32_t)a.ndc * 4096u, 0x00020000);
            // two x tiles (2 x 16 KB) in the free window buffer, two weight tiles in the ring slots the stream fills last
            // (LOOK, LOOK + 1): two steps are fetched per round trip
            const uint32_t xbuf = (uint32_t)(((par + 1) & 1) * H8_WIN), wbuf = (uint32_t)(H8_WR + LOOK * H8_TAP);
            uint32_t xo[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int q = (wave * 2 + t) * 16 + (lane >> 2);          // tile pixel of this lane's 16 bytes
                const int p = cur.p0 + q;
                const int y = (int)(((uint32_t)p * a.mg_sw) >> 20), x = cur.sx0 + (p - y * a.SW);
                const bool ok = y < a.H && x < a.W;
                const uint32_t off = ((uint32_t)((cur.img * a.ds_H + 2 * y) * a.ds_W + 2 * x) * (uint32_t)a.ds_cs + (uint32_t)a.ds_coff) * 2u +
                                     ((uint32_t)((lane & 3) ^ (((q >> 2) & 1) << 1)) << 4);
                xo[t] = ok ? off : H8_OOB;
            }
            const uint32_t dwb = (uint32_t)((2 * cb + hb) * a.ndc) * 4096u + (uint32_t)grp * 1024u;
            auto issue = [&](int d) {
                const uint32_t sel = (uint32_t)(d & 1);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rds, (lds_vp)(lds8 + xbuf + sel * 16384u + (wave * 2 + t) * 1024), 16, xo[t] + (uint32_t)d * 64u, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rdw, (lds_vp)(lds8 + wbuf + sel * H8_TAP + wave * 1024), 16, wlane, dwb + (uint32_t)d * 4096u, 0, 0);
            };
            // two steps per round trip: the gather of x[2y, 2x] (16 cache lines per 1 KiB piece) is latency, not bandwidth
            for (int d = 0; d < a.ndc; d += 2) {
                const bool two = d + 1 < a.ndc;
                issue(d);
                if (two) issue(d + 1);
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): both steps have landed (nothing else is in flight here)
                __builtin_amdgcn_s_barrier();         // ... for every wave
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 1 && !two) break;
                    const uint32_t sel = (uint32_t)h;   // issue() places step d + h in buffer (d + h) & 1 == h (d is even)
                    hvec8 wf[4], xf[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const hvec8*>(lds8 + wbuf + sel * H8_TAP + (wrd - H8_WR) + i * 1024);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int q = (grp * 4 + j) * 16 + lrow;
                        xf[j] = *reinterpret_cast<const hvec8*>(lds8 + xbuf + sel * 16384u + q * 64 + ((kg ^ (((q >> 2) & 1) << 1)) << 4));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = E::mfma(wf[i], xf[j], acc[i][j]);
                }
                if (d + 2 < a.ndc) __builtin_amdgcn_s_barrier();   // both tiles are read before the next round overwrites them
            }
            __builtin_amdgcn_s_barrier();   // the last step's tiles are read before the stream re-uses their slots
            if (MODE == 1 && hb) __builtin_amdgcn_s_barrier();
        }

        auto chunk = [&](auto last_c, const int c) {
            constexpr bool lastc = decltype(last_c)::value;
            if (lastc) {
                if (newtile && has_next) nxt = decode(ti + nslot);
                if (!has_next) nxt.y_first = a.H + 4;   // no next item: every window row is outside the image, all pieces zero-fill
                wnxt_item = wgt_base(cbn);   // no next item: re-reads weights into slots nobody reads
#pragma unroll
                for (int j = 0; j < 4; ++j) apn[j] = tap00(nxt, j);
            } else {
#pragma unroll
                for (int i = 0; i < H8_NWP; ++i) gcur[i] += 64u;   // the next chunk of the same window (H8_OOB + 64 * chunks stays out of range)
            }
            const uint32_t wthis = wcur + (uint32_t)c * H8_SLAB;
            const uint32_t wnext = lastc ? wnxt_item : wthis + H8_SLAB;
            const uint32_t winr = (uint32_t)(((par + c) & 1) * H8_WIN), winw = H8_WIN - winr;   // window buffer read / filled
            auto tap = [&](auto kk_c) {
                constexpr int kk = decltype(kk_c)::value;
                // ---------------- read segment: fragments of tap kk, DMA pieces, counted wait
                if (lastc && kk == 6) load_bias(cbn, biasn);
                if (lastc && kk == 6 && has_res) {
                    // 16-byte residual loads in the layout of the epilogue's stores.  These (8) and the bias loads (4) sit in the
                    // in-order queue between the pieces of taps 5 and 6: the waits of taps 6-8 allow that many more operations
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t ro = po[j] == H8_OOB ? H8_OOB : (po[j] * (uint32_t)a.res_cs + (uint32_t)a.res_coff) * 2u + ch0;
                        rraw[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rres, ro, 0, 0);
                        rraw[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rres, ro + 64, 0, 0);
                    }
                }
                hvec8 wf[4], xf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const hvec8*>(lds8 + wrd + kk * H8_TAP + i * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const hvec8*>(lds8 + xoff[j][kk] + winr);
                {
                    constexpr int kt = (kk + LOOK) % 9;
                    const uint32_t src = (kk + LOOK < 9 ? wthis : wnext) + (uint32_t)kt * 4096u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rwg, (lds_vp)(lds8 + H8_WR + kt * H8_TAP + wave * 1024), 16, wlane, src, 0, 0);
                }
                if constexpr (kk >= 1 && kk <= H8_NWP)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_vp)(lds8 + winw + (wave + 8 * (kk - 1)) * 1024), 16, lastc ? gnxt[kk - 1] : gcur[kk - 1], 0, 0, 0);
                constexpr bool sync_here = MODE != 2 || kk % 3 == 2;
                // an item starts drained (the epilogue's vmcnt(0)): its first taps need no wait
                if (sync_here && (c > 0 || kk >= 3)) {
                    if (lastc && kk >= 6 && has_res) h8_wait_vm<h8_allow(MODE, kk) + 12>();
                    else if (lastc && kk >= 6) h8_wait_vm<h8_allow(MODE, kk) + 4>();
                    else h8_wait_vm<h8_allow(MODE, kk)>();
                }
                if (MODE == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---------------- MFMA segment
                __builtin_amdgcn_s_setprio(1);
                if (lastc) {
                    // The next item's addresses, computed where this item no longer needs the registers (tap kk's offsets once its
                    // fragment reads are issued; window piece kk one tap before it is issued) and placed between the MFMAs: a wave
                    // issues one 16-cycle MFMA every ~4 issue slots, the address arithmetic rides in the other three
#pragma unroll
                    for (int j = 0; j < 4; ++j) xoff[j][kk] = tap_offset(apn[j], kk);
                    if constexpr (kk < H8_NWP) gnxt[kk] = win_offset(nxt, kk);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = E::mfma(wf[i], xf[j], acc[i][j]);
                if (lastc) {
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);   // up to three VALU / SALU
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                if (sync_here) {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
#ifdef ADAS_H8_PROF
                if (lastc && kk < 8) { H8P(8 + kk) }
#endif
            };
            tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); tap(std::integral_constant<int, 2>{});
            tap(std::integral_constant<int, 3>{}); tap(std::integral_constant<int, 4>{}); tap(std::integral_constant<int, 5>{});
            tap(std::integral_constant<int, 6>{}); tap(std::integral_constant<int, 7>{}); tap(std::integral_constant<int, 8>{});
#ifdef ADAS_H8_PROF
            if (lastc) { H8P(4) } else if (c == 0) { H8P(1) } else { H8P(6) }
#endif
        };
        for (int c = 0; c + 1 < a.nchunk; ++c) chunk(std::false_type{}, c);
        chunk(std::true_type{}, a.nchunk - 1);

        // ---------------- epilogue: lane holds channels kg*4..+3 of pixel lrow of every (i, j) tile (bias already in)
        H8P(2)
        // The stream's pieces, the residual and the bias retire before the stores join the queue: the counted waits of the next item
        // then never depend on how stores and loads retire relative to each other.  (The builtin form, so that hipcc's scoreboard
        // knows the residual / bias registers are complete.)
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
        // RM: residual mode as a compile-time constant (one uniform branch per item instead of selects per element)
        auto write_out = [&](auto rm_c) {
            constexpr int RM = decltype(rm_c)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t oo = po[j] == H8_OOB ? H8_OOB : (po[j] * (uint32_t)a.out_cs + (uint32_t)a.out_coff) * 2u + ch0;
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    float vx[4], vy[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vx[e] = acc[i][j][e];
                        vy[e] = acc[i + 1][j][e];
                    }
                    if (RM != RES_NONE) {
                        // v_permlane16_swap is its own inverse: the exchange that forms the 16-byte stores hands a lane its two
                        // 4-channel groups of the 16 bytes it loaded in that layout
                        const qu32x4 w = rraw[j][i >> 1];
                        const auto r0 = __builtin_amdgcn_permlane16_swap(w[0], w[2], false, false);
                        const auto r1 = __builtin_amdgcn_permlane16_swap(w[1], w[3], false, false);
                        const float rx[4] = {E::lo(r0[0]), E::hi(r0[0]), E::lo(r1[0]), E::hi(r1[0])};
                        const float ry[4] = {E::lo(r0[1]), E::hi(r0[1]), E::lo(r1[1]), E::hi(r1[1])};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            vx[e] = RM == RES_BEFORE_ACT ? h8_act<ACT>(vx[e] + rx[e]) : h8_act<ACT>(vx[e]) + rx[e];
                            vy[e] = RM == RES_BEFORE_ACT ? h8_act<ACT>(vy[e] + ry[e]) : h8_act<ACT>(vy[e]) + ry[e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            vx[e] = h8_act<ACT>(vx[e]);
                            vy[e] = h8_act<ACT>(vy[e]);
                        }
                    }
                    const uint32_t x0 = E::pack2(vx[0], vx[1]), x1 = E::pack2(vx[2], vx[3]);
                    const uint32_t y0 = E::pack2(vy[0], vy[1]), y1 = E::pack2(vy[2], vy[3]);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                    __builtin_amdgcn_raw_buffer_store_b128(qu32x4{s0[0], s1[0], s0[1], s1[1]}, rout, oo + i * 32, 0, 0);
                    // the next item's accumulators start at its bias
                    acc[i][j] = qf32x4{biasn[i].x, biasn[i].y, biasn[i].z, biasn[i].w};
                    acc[i + 1][j] = qf32x4{biasn[i + 1].x, biasn[i + 1].y, biasn[i + 1].z, biasn[i + 1].w};
                }
            }
        };
        if (a.res_mode == RES_NONE) write_out(std::integral_constant<int, RES_NONE>{});
        else if (a.res_mode == RES_BEFORE_ACT) write_out(std::integral_constant<int, RES_BEFORE_ACT>{});
        else write_out(std::integral_constant<int, RES_AFTER_ACT>{});

        H8P(3)
#ifdef ADAS_H8_PROF
        ++nit__;
#endif
        if (!has_next) break;
        if (newtile) {
            ti += nslot;
            cur = nxt;
#pragma unroll
            for (int j = 0; j < 4; ++j) po[j] = out_pixel(cur, j);
        }
        cb = cbn;
        cbi = newtile ? 0 : cbi + 1;
#pragma unroll
        for (int i = 0; i < H8_NWP; ++i) gcur[i] = gnxt[i];
        wcur = wnxt_item;
        par = (par + a.nchunk) & 1;
    }
    if (MODE == 1 && !hb) __builtin_amdgcn_s_barrier();   // pairs with group 1's extra barrier
#ifdef ADAS_H8_PROF
    if (lane == 0 && grp == 0) {
        unsigned long long* b__ = g_h8_prof[blockIdx.x & 255];
        for (int i__ = 0; i__ < 16; ++i__)
            if (i__ != 7) atomicAdd(&b__[i__ + 16 * hb], pacc__[i__]);
        atomicAdd(&b__[7 + 16 * hb], (unsigned long long)nit__);
    }
#endif
}

#ifdef ADAS_H8_PROF
extern "C" int adas_debug_h8_prof(unsigned long long* out16, int reset) {
    static unsigned long long h[256][32];
    if (out16) {   // 32 values: 16 per wave group
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_h8_prof), sizeof(h)) != hipSuccess) return -1;
        for (int i = 0; i < 32; ++i) {
            out16[i] = 0;
            for (int b = 0; b < 256; ++b) out16[i] += h[b][i];
        }
    }
    if (reset) {
        memset(h, 0, sizeof(h));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_h8_prof), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// -------------------------------------------------------------------------------------
static int h8_mode() {   // ADAS_HALO8: 0 off, 1 on (default: synchronisation variant picked per layer), 2 / 3 force variant 2 / 1
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_HALO8");
        v = e ? atoi(e) : 1;
    }
    return v;
}

// channel blocks a workgroup computes back to back on one tile: the largest divisor of ncb that still leaves every CU of an XCD
// at least one unit and wastes under a fifth of the last round (0: none does)
static int h8_blocks_per_unit(long tiles8, int ncb) {
    for (int cpw = ncb; cpw >= 1; --cpw) {
        if (ncb % cpw) continue;
        const long units8 = tiles8 * (ncb / cpw), rounds = (units8 + H8_SLOTS - 1) / H8_SLOTS;
        if (units8 >= H8_SLOTS && (double)units8 / (double)(rounds * H8_SLOTS) >= 0.8) return cpw;
    }
    return 0;
}

bool halo8_applicable(int kh, int kw, int stride, int pad, int n, const TView& in, const TView& out, const TView& res, int res_mode) {
    if (!h8_mode()) return false;
    if (kh != 3 || kw != 3 || stride != 1 || pad != 1) return false;
    if (in.f32 || out.f32 || out.h != in.h || out.w != in.w) return false;
    if ((out.c & 127) || (in.c & 31) || in.c < 64) return false;
    if ((in.cs & 7) || (in.coff & 7) || (out.cs & 7) || (out.coff & 7)) return false;
    if (res_mode != RES_NONE && ((res.cs & 7) || (res.coff & 7) || res.f32)) return false;
    if ((double)n * in.h * in.w * in.cs * 2.0 >= (double)H8_OOB || (double)n * out.h * out.w * out.cs * 2.0 >= (double)H8_OOB) return false;
    if (res_mode != RES_NONE && (double)n * out.h * out.w * res.cs * 2.0 >= (double)H8_OOB) return false;
    HaloPlan pl;
    if (!plan_halo(out.h, out.w, 1, &pl, H8_MAXPIX, 0, h8_policy()) || pl.eff < 0.6 || pl.maxpix > H8_MAXPIX) return false;
    // one workgroup per CU walking its XCD's items in rounds of 32: the launch has to fill the chip, in nearly whole rounds
    const long ntiles = (long)n * pl.NS * pl.TPS, tiles8 = (ntiles + 7) / 8;
    if (ntiles * (out.c / 128) * pl.NS * pl.TPS >= (1L << 32)) return false;
    return h8_blocks_per_unit(tiles8, out.c / 128) > 0;
}

static bool h8_ds_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_NO_DS_FUSE");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

bool halo8_ds_applicable(int kh, int kw, int stride, int pad, int n, const TView& in, const TView& out, const TView& x) {
    if (!h8_ds_enabled()) return false;
    TView none = out;
    none.p = nullptr;
    if (!halo8_applicable(kh, kw, stride, pad, n, in, out, none, RES_NONE)) return false;
    if (x.f32 || (x.c & 31) || x.c < 32 || x.c > 512 || ((x.cs | x.coff) & 7)) return false;
    if ((x.h + 1) / 2 != out.h || (x.w + 1) / 2 != out.w) return false;   // 1x1, stride 2, pad 0
    return (double)n * x.h * x.w * x.cs * 2.0 < (double)H8_OOB;
}

__device__ __forceinline__ void h8_store(uint16_t* p, float v) { *p = Bf16::from_f32(v); }
__device__ __forceinline__ void h8_store(f16s* p, float v) { p->v = Fp16::from_f32(v); }

// fp32 [cout][cin] -> 16-bit [cout / 64][cin / 32][64 rows][32]: the per-step weight tiles of the projection pre-pass
template <typename T>
__global__ void pack_weights_ds_kernel(const float* __restrict__ src, T* __restrict__ dst, int cout, int cin) {
    const int total = cout * cin;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i & 31;
        int r = i >> 5;
        const int row = r & 63; r >>= 6;
        const int nd = cin >> 5;
        const int d = r % nd, tile = r / nd;
        const float v = src[(size_t)(tile * 64 + row) * cin + d * 32 + c];
        h8_store(dst + i, v);
    }
}
hipError_t launch_pack_weights_ds(const float* src, void* dst, int cout, int cin, int prec, hipStream_t st) {
    if ((cout & 63) || (cin & 31)) return hipErrorInvalidValue;
    const int blocks = (cout * cin + 255) / 256;
    if (prec == PREC_FP16) hipLaunchKernelGGL(pack_weights_ds_kernel<f16s>, dim3(blocks), dim3(256), 0, st, src, (f16s*)dst, cout, cin);
    else hipLaunchKernelGGL(pack_weights_ds_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st, src, (uint16_t*)dst, cout, cin);
    return hipGetLastError();
}

template <typename E, int MODE>
static hipError_t h8_launch(const H8Dev& d, int act, dim3 grid, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_h8_kernel<E, ACT_NONE, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_h8_kernel<E, ACT_SILU, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_h8_kernel<E, ACT_RELU, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_h8_kernel<E, ACT_LEAKY, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    if (act == ACT_SILU) hipLaunchKernelGGL((conv_h8_kernel<E, ACT_SILU, MODE>), grid, dim3(H8_THR), H8_LDS, st, d);
    else if (act == ACT_RELU) hipLaunchKernelGGL((conv_h8_kernel<E, ACT_RELU, MODE>), grid, dim3(H8_THR), H8_LDS, st, d);
    else if (act == ACT_LEAKY) hipLaunchKernelGGL((conv_h8_kernel<E, ACT_LEAKY, MODE>), grid, dim3(H8_THR), H8_LDS, st, d);
    else hipLaunchKernelGGL((conv_h8_kernel<E, ACT_NONE, MODE>), grid, dim3(H8_THR), H8_LDS, st, d);
    return hipGetLastError();
}

hipError_t launch_conv_halo8(const ConvArgs& a, hipStream_t st) {
    HaloPlan pl;
    if (!halo8_applicable(a.kh, a.kw, a.stride, a.pad, a.n, a.in, a.out, a.res, a.res_mode) || !plan_halo(a.out.h, a.out.w, 1, &pl, H8_MAXPIX, 0, h8_policy()))
        return hipErrorNotSupported;
    {   // ADAS_H8_SW=<strip width>: narrower strips = squarer tiles = less halo per window (and more padded pixels): an experiment knob
        static int sw = -1;
        if (sw < 0) { const char* e = getenv("ADAS_H8_SW"); sw = e ? atoi(e) : 0; }
        HaloPlan alt;
        if (sw > 0 && plan_halo_sw(a.out.h, a.out.w, 1, sw, H8_MAXPIX, &alt) && alt.eff >= 0.6) pl = alt;
    }
    H8Dev d;
    d.in = (const uint16_t*)a.in.p; d.wgt = (const uint16_t*)a.wgt; d.bias = a.bias; d.out = (uint16_t*)a.out.p;
    d.res = (const uint16_t*)a.res.p;
    d.in_cs = a.in.cs; d.in_coff = a.in.coff; d.cin = a.in.c; d.H = a.in.h; d.W = a.in.w;
    d.out_cs = a.out.cs; d.out_coff = a.out.coff; d.cout = a.out.c;
    d.res_cs = a.res.cs; d.res_coff = a.res.coff; d.res_mode = a.res_mode;
    d.nchunk = (a.in.c + 31) / 32;
    d.in_bytes = (uint32_t)((size_t)a.n * a.in.h * a.in.w * a.in.cs * 2);
    d.wgt_bytes = (uint32_t)((size_t)(a.out.c / 64) * d.nchunk * H8_SLAB);
    d.out_bytes = (uint32_t)((size_t)a.n * a.out.h * a.out.w * a.out.cs * 2);
    d.res_bytes = a.res_mode != RES_NONE ? (uint32_t)((size_t)a.n * a.out.h * a.out.w * a.res.cs * 2) : 0u;
    d.SW = pl.SW; d.NS = pl.NS; d.TPS = pl.TPS; d.WW = pl.WW;
    d.mg_ww = pl.mg_ww; d.mg_sw = pl.mg_sw;
    d.mg_img = (uint32_t)(((1ull << 32) + (uint64_t)(pl.NS * pl.TPS) - 1) / (uint64_t)(pl.NS * pl.TPS));   // unused when the divisor is 1
    d.mg_tps = (uint32_t)(((1ull << 32) + (uint64_t)pl.TPS - 1) / (uint64_t)pl.TPS);

    d.ntiles = a.n * pl.NS * pl.TPS;
    d.tiles8 = (d.ntiles + 7) / 8;
    d.ncb = a.out.c / 128;
    d.ds_in = nullptr; d.ds_w = nullptr; d.ds_bias = nullptr; d.ds_bytes = 0; d.ds_cs = d.ds_coff = d.ds_H = d.ds_W = d.ndc = 0;
    if (a.ds_w) {
        d.ds_in = (const uint16_t*)a.ds_in.p; d.ds_w = (const uint16_t*)a.ds_w; d.ds_bias = a.ds_bias;
        d.ds_bytes = (uint32_t)((size_t)a.n * a.ds_in.h * a.ds_in.w * a.ds_in.cs * 2);
        d.ds_cs = a.ds_in.cs; d.ds_coff = a.ds_in.coff; d.ds_H = a.ds_in.h; d.ds_W = a.ds_in.w; d.ndc = a.ds_in.c / 32;
        d.res_mode = RES_NONE;   // the shortcut arrives through the MFMAs
    }
    d.cpw = h8_blocks_per_unit(d.tiles8, d.ncb);
    if (d.cpw <= 0) return hipErrorNotSupported;
    const int upt = d.ncb / d.cpw;
    d.mg_upt = (uint32_t)(((1ull << 32) + (uint64_t)upt - 1) / (uint64_t)upt);
    const int units8 = d.tiles8 * upt;
    const int cap_slots = persist_slots(0);
    const int slots = units8 < cap_slots ? units8 : cap_slots;
    dim3 grid(8 * slots);
    const int forced = h8_mode();
    const bool pingpong = forced == 3 || (forced != 2 && d.nchunk >= 12);
    if (a.prec == PREC_FP16) return pingpong ? h8_launch<Fp16, 1>(d, a.act, grid, st) : h8_launch<Fp16, 2>(d, a.act, grid, st);
    return pingpong ? h8_launch<Bf16, 1>(d, a.act, grid, st) : h8_launch<Bf16, 2>(d, a.act, grid, st);
}

}  // namespace adas
