// conv_halo8_x3.hip -- the stride-1 3x3 convolution of the split precision (ADAS_PREC_FP16X3) on conv_halo8.hip's structure: one
// persistent 8-wave workgroup per CU, window and per-tap weight tiles fed by LDS-DMA, counted waits, never draining inside the stream.
//
// conv_x3.hip (the generic kernel) gathers every input pixel once per tap from L2: at three MFMAs per product it is L2-bound at
// ~500 TFLOP/s of MFMA work (measured: 150-207 TFLOP/s of conv work on the UFLD layers).  Here the split operands run through the
// halo stream as HALF-CHUNKS: a 32-channel chunk of the G8 activation tensor (128 B per pixel: four groups of [16 B hi | 16 B lo])
// is streamed as an H chunk (the four hi pieces) followed by an L chunk (the four lo pieces), each through the same 64-byte window
// pixels, swizzle and double buffering as the 16-bit kernel -- the DMA's per-lane source address picks hi or lo.
//   item      256 pixels x 64 output channels; wave group hb owns 32 of them as four 16-row MFMA tiles:
//             rows 0-31 = MAIN (w_hi), rows 32-63 = CROSS (H chunk: w_lo * 2^11; L chunk: w_hi)
//   H chunk   main += w_hi a_hi (8 MFMAs per wave and tap), cross += w_lo a_hi (8)
//   L chunk   cross += w_hi a_lo (8); the main rows of its weight tiles are never read
//   epilogue  out = act(main + 2^-11 cross [+ residual]) -> split -> G8 store (8 B hi + 8 B lo per lane and 16-channel tile)
// = the three-MFMA product of elem16.h with no operand re-fetch.  Weight slabs [64-row block][half-chunk][tap][64 rows][32] are packed at
// load time (launch_pack_weights_h8x3); an eligible conv keeps both this packing and the generic one, the launch picks by batch.
#include "kernels.h"
#include "elem16.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float yf32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t yu32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t yu32x2;
typedef __attribute__((address_space(3))) void* ylds_vp;

struct H8XDev {
    const void* in;
    const void* wgt;
    const float* bias;
    void* out;
    const void* res;
    uint32_t in_bytes, wgt_bytes, out_bytes, res_bytes;
    int in_cs, in_coff, cin, H, W;
    int out_cs, out_coff, cout;
    int res_cs, res_coff, res_mode;
    int nck;                  // half-chunks: 2 * (cin / 32), order H0 L0 H1 L1 ...
    int SW, NS, TPS, WW;      // strip width, strips per row, tiles per strip, window width (conv_halo.hip's plan)
    uint32_t mg_ww, mg_sw;
    uint32_t mg_img, mg_tps, mg_upt;
    int ntiles, tiles8, ncb, cpw;  // tiles, ceil(tiles / 8), 64-channel blocks, blocks per unit
};

constexpr int X8_THR = 512;
constexpr int X8_BM = 256;
constexpr int X8_MAXPIX = 640;
constexpr int X8_WIN = X8_MAXPIX * 64;          // bytes of one window buffer (one half-chunk: 32 halves per pixel)
constexpr int X8_TAP = 2 * 64 * 64;             // bytes of one tap's weights (two 64-row blocks)
constexpr int X8_WR = 2 * X8_WIN;               // byte offset of the weight ring
constexpr int X8_LDS = X8_WR + 9 * X8_TAP;      // 155,648 B
constexpr int X8_NWP = X8_MAXPIX / 16 / 8;      // window pieces per wave (5)
constexpr int X8_SLAB = 9 * 64 * 64;            // bytes of one (64-row block, half-chunk) weight slab
constexpr uint32_t X8_OOB = 0xF0000000u;
constexpr int X8_SLOTS = 32;

__host__ __device__ constexpr int x8_look(int mode) { return mode == 2 ? 6 : 4; }

template <int N>
__device__ __forceinline__ void x8_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int ACT>
__device__ __forceinline__ float x8_act(float v) {
    if (ACT == ACT_SILU) return x3_silu(v);   // (elem16.h: fp32-class, 12 instructions)
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_LEAKY) return fmaxf(v, 0.1f * v);
    return v;
}

// ---- the epilogue's arithmetic two values at a time (round 6; ADAS_H8X_SCALAR_EPI builds keep the one-value form).  An item's epilogue was
// ~16 VALU instructions per output value, 32 values per lane, two waves per SIMD: ~4,000 cycles in which no MFMA issues.  gfx950 has
// v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 and v_cvt_pk_f16_f32: join, residual add and split take 12 instructions per four values
// instead of 28 + 12.  x3_split's "no subnormal hi" rule costs a compare and a select per value in the scalar form; here the hi
// conversions of a whole item run with MODE.FP_DENORM[7:6] = 0 (16-bit results flushed: a hi below the half normal range becomes 0
// and the value moves into lo, the same rule) between two s_setreg -- the conversions are volatile asm so that they stay between them.
typedef __attribute__((ext_vector_type(2))) float yf32x2;
template <int ACT>
__device__ __forceinline__ yf32x2 x8_act2(yf32x2 v) {
    if (ACT == ACT_RELU) return __builtin_elementwise_max(v, yf32x2{0.0f, 0.0f});
    return yf32x2{x8_act<ACT>(v[0]), x8_act<ACT>(v[1])};
}
__device__ __forceinline__ yf32x2 x8_join2(uint32_t h, uint32_t l) {   // two (hi, lo) pairs -> their values
    return __builtin_convertvector(__builtin_bit_cast(e_f16x2, l), yf32x2) * kX3Down + __builtin_convertvector(__builtin_bit_cast(e_f16x2, h), yf32x2);
}
__device__ __forceinline__ void x8_hi_mode(bool flush) {   // MODE[7:6] (FP_DENORM of 16- and 64-bit results): 0 = flush, 3 = the launch default
    if (flush) __builtin_amdgcn_s_setreg((unsigned short)(1 | (6 << 6) | (1 << 11)), 0u);
    else __builtin_amdgcn_s_setreg((unsigned short)(1 | (6 << 6) | (1 << 11)), 3u);
}
__device__ __forceinline__ uint32_t x8_cvt_hi2(yf32x2 v) {
    uint32_t h;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(v[0]), "v"(v[1]));
    return h;
}
__device__ __forceinline__ uint32_t x8_lo2(yf32x2 v, uint32_t h) {   // the lo halves of two values given their hi halves
    const yf32x2 d = (v - __builtin_convertvector(__builtin_bit_cast(e_f16x2, h), yf32x2)) * kX3Up;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(d, e_f16x2));
}

__host__ __device__ constexpr int x8_issued(int k) { return 1 + ((k >= 1 && k <= X8_NWP) ? 1 : 0); }
__host__ __device__ constexpr int x8_allow(int mode, int k) {
    return mode == 2 ? x8_issued(k) + x8_issued(k - 1) + x8_issued(k - 2)
                     : x8_issued(k) + x8_issued((k + 8) % 9) + x8_issued((k + 7) % 9);
}
// loads queued under the last tap row of an item besides the stream's pieces: the next item's bias (2 float4) and, with a residual,
// 16 eight-byte loads (4 pixel tiles x 2 channel tiles x {hi, lo})
// X8_RES_TAP: the tap of an item's LAST half-chunk under which the residual tile and the next item's bias are requested.  Round 4 asked
// at tap 6 (the last tap row), which leaves an HBM read ~1.5 k cycles before the epilogue needs it -- the phase counters showed the last
// half-chunk of a residual layer at 11.9 k cycles against 6.5 k without (profiles/r06/h8x_phases.txt); at tap 0 the loads have the whole
// half-chunk (the in-order return then asks them to be back by the wait of tap 5, ~3 k cycles on).  Same registers either way; measured
// +0.2 % end to end (inside the noise: the item's other fixed costs hide most of it), kept.  (Later in round 6: one residual load per tap,
// taps 0-7, instead of eight under tap 0 -- counted waits adjusted per tap row -- changed nothing: layer1's conv2 0.515 / 0.509 ms either
// way, -0.3 % end to end; the 3 k cycles a residual item costs are its 64 KB of extra HBM reads, not the issue burst.  Not kept.)
#ifndef ADAS_H8X_RES_TAP
#define ADAS_H8X_RES_TAP 0
#endif
constexpr int X8_RES_TAP = ADAS_H8X_RES_TAP;
#ifdef ADAS_H8X_NARROW_ST
constexpr int X8_NBIAS = 2, X8_NRES = 16;
#else
constexpr int X8_NBIAS = 2, X8_NRES = 8;    // (wide form: one 16-byte load per pixel tile and channel tile)
#endif

#ifdef ADAS_H8X_PROF   // scratch instrumentation (tools/experiments/h8x_prof.py): shader cycles of waves 0 and 4 per item phase
__device__ unsigned long long g_h8x_prof[256][32];
#define H8XP(i)                                     \
    if (lane == 0 && grp == 0) {                    \
        const unsigned long long t__ = clock64();   \
        pacc__[i] += t__ - tprev__;                 \
        tprev__ = t__;                              \
    }
#else
#define H8XP(i)
#endif

// LH: the weight tiles of an L half-chunk are fetched as their CROSS rows only (rows 32-63 of each 64-row block: the main rows are
// zeros that no MFMA reads) -- every wave still issues one piece per tap, with the upper half of its lanes masked off: 512 B = 8 rows
// SH (round 6, MODE 2 only): the L half-chunk of a 32-channel chunk multiplies the lo activations by w_hi -- the MAIN rows of the H
// half-chunk's own weight tiles, which sit in the 9-slot ring while the H taps run.  With SH the ring keeps a chunk's nine tiles through
// BOTH half-chunks and the L taps read those main rows: no weight tile is fetched for an L half-chunk at all (half the weight DMA
// instructions and bytes of the stream).  Tile t of the next chunk may overwrite slot t once the L tap row that reads it has passed its
// barrier, and must be issued two tap rows before the H tap row that reads it: exactly one tap row fits -- tiles 0-2 go out under L
// taps 3-5, tiles 3-5 under L taps 6-8, tiles 6-8 under the next H taps 0-2 ("tile (kk + 6) % 9" as before, on those taps only).
// NWP: window pieces per wave and half-chunk (1 KiB each, 16 pixels): 3 / 4 / 5 for windows up to 384 / 512 / 640 pixels -- a tile's
// window rarely fills the 640-pixel buffer, and every piece is a DMA instruction whether its lanes are in range or not.
__host__ __device__ constexpr int x8s_issued(bool lo, int k, int nwp) { return ((lo ? k >= 3 : (k >= 0 && k < 3)) ? 1 : 0) + ((k >= 1 && k <= nwp) ? 1 : 0); }
__host__ __device__ constexpr int x8s_allow(bool lo, int k, int nwp) { return x8s_issued(lo, k, nwp) + x8s_issued(lo, k - 1, nwp) + x8s_issued(lo, k - 2, nwp); }

template <int ACT, int MODE, bool LH, bool SH, int NWP>
__global__ __launch_bounds__(X8_THR, 1) void conv_h8x3_kernel(H8XDev a) {
    static_assert(!SH || (MODE == 2 && !LH), "shared weight tiles: one barrier per tap row, whole tiles");
    static_assert(NWP >= 1 && NWP <= X8_NWP && (SH || NWP == X8_NWP), "window pieces per wave");
    Fp16::enter();
    typedef Fp16::vec8 hvec8;
    constexpr int LOOK = x8_look(MODE);
    extern __shared__ __attribute__((aligned(16))) uint8_t lds8[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, kg = lane >> 4;
    const int hb = wave >> 2, grp = wave & 3;   // 32-channel half of the item (= wave group), pixel quarter
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
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
    const uint32_t wlane = (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ gsw[(lane >> 4) & 3]) << 4));
    // window: LDS position (lane & 3) of pixel (lane >> 2) holds K group (lane & 3) ^ swizzle; in the G8 tensor that group's hi
    // halves sit at group * 32 bytes (its lo halves 16 bytes behind)
    const uint32_t wpiece = (uint32_t)((lane & 3) ^ (((lane >> 4) & 1) << 1)) << 5;
    const uint32_t wrd = (uint32_t)(X8_WR + hb * 4096 + lrow * 64 + ((kg ^ gsw[(lrow >> 2) & 3]) << 4));
    // LH pieces: wave grp brings rows 32 + 8 grp .. + 7 of its block (lanes 0-31: row lane >> 2), i.e. rows 8 (grp & 1) .. + 7 of the
    // 16-row tile 2 + (grp >> 1) -- the swizzle key is the row's position in THAT tile
    const uint32_t wlane_h = (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ gsw[(((lane >> 4) & 1) + 2 * (grp & 1)) & 3]) << 4));
    const uint32_t wdst_h = (uint32_t)(X8_WR + hb * 4096 + 2048 + grp * 512);
    // epilogue: lane owns channels kg*4 .. +3 of 16-channel tile i of the wave group's 32: G8 byte offset of their hi halves within
    // the pixel's 64-channel run (lo halves 16 bytes behind): group (hb*32 + i*16 + kg*4) / 8, element (kg & 1) * 4
    // Wide form (default): v_permlane16_swap pairs the lanes that own the two 4-channel halves of a group (kg even / odd: 16 lanes
    // apart), after which the even row holds the group's 16 hi bytes and the odd row its 16 lo bytes: one 16-byte store (and residual
    // load) per lane and tile instead of two 8-byte ones -- half the vector-memory instructions of the epilogue.
#ifdef ADAS_H8X_NARROW_ST
    const uint32_t ch_lane = (uint32_t)((hb * 4 + (kg >> 1)) * 32 + (kg & 1) * 8);
#else
    const uint32_t ch_lane = (uint32_t)((hb * 4 + (kg >> 1)) * 32 + (kg & 1) * 16);
#endif
    const bool has_res = a.res_mode != RES_NONE;

    struct Tile {
        int img, sx0, p0, y_first;
    };
    auto decode = [&](int u) {
        Tile t;
        int tile = xcd * a.tiles8 + (upt == 1 ? u : (int)__umulhi((uint32_t)u, a.mg_upt));
        t.img = per_img == 1 ? tile : (int)__umulhi((uint32_t)tile, a.mg_img);
        tile -= t.img * per_img;
        const int strip = a.TPS == 1 ? tile : (int)__umulhi((uint32_t)tile, a.mg_tps);
        t.sx0 = strip * a.SW;
        t.p0 = (tile - strip * a.TPS) * X8_BM;
        t.y_first = (int)(((uint32_t)t.p0 * a.mg_sw) >> 20);
        return t;
    };
    // source byte offset of this lane's 16 bytes in the wave's window piece i (half-chunk 0 = hi of channels 0-31)
    auto win_offset = [&](const Tile& t, int i) {
        const int y_lastp = (int)(((uint32_t)(t.p0 + X8_BM - 1) * a.mg_sw) >> 20);
        const int npix = (y_lastp - t.y_first + 3) * a.WW;
        const int pix = (wave + 8 * i) * 16 + (lane >> 2);
        const int wy = (int)(((uint32_t)pix * a.mg_ww) >> 20), wx = pix - wy * a.WW;
        const int iy = t.y_first - 1 + wy, ix = t.sx0 - 1 + wx;
        const bool ok = pix < npix && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const uint32_t off = ((uint32_t)((t.img * a.H + iy) * a.W + ix) * (uint32_t)a.in_cs + (uint32_t)a.in_coff) * 4u + wpiece;
        const uint32_t m = 0u - (uint32_t)ok;
        return (off & m) | (X8_OOB & ~m);
    };
    auto tap00 = [&](const Tile& t, int j) {
        const int p = t.p0 + (grp * 4 + j) * 16 + lrow;
        const int y = (int)(((uint32_t)p * a.mg_sw) >> 20), xs = p - y * a.SW;
        return (uint32_t)((((y - t.y_first) * a.WW + xs) << 6) | (kg << 4));
    };
    auto tap_offset = [&](uint32_t ap64, int t) {
        const uint32_t v = ap64 + (uint32_t)(((t / 3) * a.WW + (t % 3)) << 6);
        return v ^ ((v >> 3) & 0x20u);
    };
    auto out_pixel = [&](const Tile& t, int j) {
        const int p = t.p0 + (grp * 4 + j) * 16 + lrow;
        const int y = (int)(((uint32_t)p * a.mg_sw) >> 20), x = t.sx0 + (p - y * a.SW);
        return (y < a.H && x < a.W) ? (uint32_t)((t.img * a.H + y) * a.W + x) : X8_OOB;
    };
    // scalar byte offset of this wave's 16 rows of (64-channel block cb, half hb, half-chunk 0, tap 0)
    auto wgt_base = [&](int cb) { return (uint32_t)(((2 * cb + hb) * a.nck) * X8_SLAB + grp * 1024); };
    auto load_bias = [&](int cb, float4* b) {
#pragma unroll
        for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const float4*>(a.bias + cb * 64 + hb * 32 + i * 16 + kg * 4);
    };

    auto first_cb = [&](int u) { return upt == 1 ? 0 : (u - (int)__umulhi((uint32_t)u, a.mg_upt) * upt) * a.cpw; };
    int ti = slot, cb = first_cb(slot), cbi = 0;
    Tile cur = decode(ti);
    uint32_t gcur[X8_NWP], gnxt[X8_NWP];   // (the first NWP entries are used; a template-sized array here makes hipcc drop the host stub)
    uint32_t xoff[4][9];
    uint32_t po[4];
#pragma unroll
    for (int i = 0; i < NWP; ++i) gcur[i] = win_offset(cur, i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t ap64 = tap00(cur, j);
#pragma unroll
        for (int t = 0; t < 9; ++t) xoff[j][t] = tap_offset(ap64, t);
        po[j] = out_pixel(cur, j);
    }
    uint32_t wcur = wgt_base(cb);
    int par = 0;
    float4 biasn[2];
    load_bias(cb, biasn);

    // ---- prologue: window of half-chunk 0 into buffer 0, weights of taps 0 .. LOOK-1
#pragma unroll
    for (int i = 0; i < NWP; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (ylds_vp)(lds8 + (wave + 8 * i) * 1024), 16, gcur[i], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < LOOK; ++t)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwg, (ylds_vp)(lds8 + X8_WR + t * X8_TAP + wave * 1024), 16, wlane, wcur + t * 4096, 0, 0);
    yf32x4 acc[4][4];   // [0..1]: main, starts at the bias; [2..3]: cross, starts at zero
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            acc[i][j] = yf32x4{biasn[i].x, biasn[i].y, biasn[i].z, biasn[i].w};
            acc[i + 2][j] = yf32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    x8_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (MODE == 1 && hb) __builtin_amdgcn_s_barrier();   // group 1 runs one segment behind group 0
#ifdef ADAS_H8X_PROF
    unsigned long long tprev__ = clock64();
    unsigned long long pacc__[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int nit__ = 0;
#endif

    for (;;) {
        H8XP(0)
        const bool newtile = cbi + 1 == a.cpw;
        const bool has_next = !newtile || ti + nslot < units_here;
        const int cbn = newtile ? first_cb(has_next ? ti + nslot : ti) : cb + 1;
        const uint32_t ch0 = (uint32_t)(cb * 256) + ch_lane;   // 64 channels = 256 bytes of G8 storage
        // a wave group whose 32 channels lie past Cout (the padded tail of the last 64-channel block: 80 -> 128 leaves group 1 of block 1 with
        // zero weight rows and masked stores) takes part in the DMA stream and the barriers but reads no fragments and issues no MFMAs -- its
        // SIMD partner of group 0 has the matrix pipe to itself for that item (ADAS_H8X_NO_TRIM builds keep the zero products)
#ifndef ADAS_H8X_NO_TRIM
        const bool wave_on = cb * 64 + hb * 32 < a.cout;
#else
        constexpr bool wave_on = true;
#endif
        Tile nxt = cur;
        uint32_t wnxt_item = 0, apn[4] = {0, 0, 0, 0};
        yu32x4 rraw[4][2];      // residual [pixel tile][channel tile], fetched under the last row of taps of the last half-chunk: one 16-byte piece
                                // of the lane's G8 group per entry (wide form), or its own 8 B hi + 8 B lo (ADAS_H8X_NARROW_ST builds)

        auto chunk = [&](auto last_c, auto lo_c, const int c) {
            constexpr bool lastc = decltype(last_c)::value;
            constexpr bool islo = decltype(lo_c)::value;   // L half-chunk: only the cross tiles (i = 2, 3) accumulate
            if (lastc) {
                if (newtile && has_next) nxt = decode(ti + nslot);
                if (!has_next) nxt.y_first = a.H + 4;
                wnxt_item = wgt_base(cbn);
#pragma unroll
                for (int j = 0; j < 4; ++j) apn[j] = tap00(nxt, j);
            } else {
                // the half-chunk fetched during this one: after an H chunk its own lo pieces (16 bytes on), after an L chunk the hi
                // pieces of the next 32 channels (a group is 32 bytes, a chunk 128)
#pragma unroll
                for (int i = 0; i < NWP; ++i) gcur[i] += islo ? 112u : 16u;
            }
            const uint32_t wthis = wcur + (uint32_t)c * X8_SLAB;
            const uint32_t wnext = lastc ? wnxt_item : wthis + X8_SLAB;
            const uint32_t winr = (uint32_t)(((par + c) & 1) * X8_WIN), winw = X8_WIN - winr;
            auto tap = [&](auto kk_c) {
                constexpr int kk = decltype(kk_c)::value;
                if (lastc && kk == X8_RES_TAP) load_bias(cbn, biasn);
                if (lastc && kk == X8_RES_TAP && has_res) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t ro = po[j] == X8_OOB ? X8_OOB : (po[j] * (uint32_t)a.res_cs + (uint32_t)a.res_coff) * 4u + ch0;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
#ifdef ADAS_H8X_NARROW_ST
                            const uint32_t ri = (cb * 64 + hb * 32 + i * 16 + kg * 4 < a.cout) ? ro + i * 64 : X8_OOB;
                            const yu32x2 rh = __builtin_bit_cast(yu32x2, __builtin_amdgcn_raw_buffer_load_b64(rres, ri, 0, 0));
                            const yu32x2 rl = __builtin_bit_cast(yu32x2, __builtin_amdgcn_raw_buffer_load_b64(rres, ri + 16, 0, 0));
                            rraw[j][i] = yu32x4{rh.x, rh.y, rl.x, rl.y};
#else
                            // one 16-byte piece of the lane's 8-channel group: its hi half on the even 16-lane rows, its lo half on the odd ones
                            const uint32_t ri = (cb * 64 + hb * 32 + i * 16 + (kg >> 1) * 8 < a.cout) ? ro + i * 64 : X8_OOB;
                            rraw[j][i] = __builtin_bit_cast(yu32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, ri, 0, 0));
#endif
                        }
                    }
                }
                hvec8 wf[4], xf[4];
                // (SH: an L tap multiplies by the MAIN rows -- w_hi -- of the chunk's H tile, still in slot kk)
                if (wave_on) {
#pragma unroll
                    for (int i = islo ? 2 : 0; i < 4; ++i) wf[i] = *reinterpret_cast<const hvec8*>(lds8 + wrd + kk * X8_TAP + ((SH && islo) ? i - 2 : i) * 1024);
#pragma unroll
                    for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const hvec8*>(lds8 + xoff[j][kk] + winr);
                }
                if constexpr (SH) {
                    if constexpr (islo ? kk >= 3 : kk < 3) {
                        constexpr int kt = (kk + 6) % 9;      // H taps 0-2: tiles 6-8 of this chunk; L taps 3-8: tiles 0-5 of the next chunk (or item)
                        const uint32_t src = (islo ? wnext : wthis) + (uint32_t)kt * 4096u;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwg, (ylds_vp)(lds8 + X8_WR + kt * X8_TAP + wave * 1024), 16, wlane, src, 0, 0);
                    }
                } else {
                    constexpr int kt = (kk + LOOK) % 9;
                    constexpr bool tgt_lo = (kk + LOOK < 9) ? islo : !islo;     // the half-chunk this tap tile belongs to (the chunks alternate H, L)
                    const uint32_t src = (kk + LOOK < 9 ? wthis : wnext) + (uint32_t)kt * 4096u;
                    if (LH && tgt_lo) {
                        // (wthis / wnext carry this wave's grp * 1024: the cross rows of the wave's piece sit at + 2048 - 512 grp from there)
                        if (lane < 32)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwg, (ylds_vp)(lds8 + wdst_h + kt * X8_TAP), 16, wlane_h, src + 2048u - (uint32_t)grp * 512u, 0, 0);
                    } else {
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwg, (ylds_vp)(lds8 + X8_WR + kt * X8_TAP + wave * 1024), 16, wlane, src, 0, 0);
                    }
                }
                if constexpr (kk >= 1 && kk <= NWP)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (ylds_vp)(lds8 + winw + (wave + 8 * (kk - 1)) * 1024), 16, lastc ? gnxt[kk - 1] : gcur[kk - 1], 0, 0, 0);
                constexpr bool sync_here = MODE != 2 || kk % 3 == 2;
                // what may stay in flight past this point: the pieces issued in this tap row (MODE 2) / the last three taps
                constexpr int allow = SH ? x8s_allow(islo, kk, NWP) : x8_allow(MODE, kk);
                // (the residual / bias loads of the last half-chunk count among the pieces of the tap row they are issued in -- MODE 2 -- or of the
                // following taps -- MODE 1, one wait per tap: X8_RES_TAP = 6 there)
                constexpr bool with_res = kk >= X8_RES_TAP && (MODE == 2 ? kk / 3 == X8_RES_TAP / 3 : kk < X8_RES_TAP + 3);
                auto counted_wait = [&]() {
                    if (sync_here && (c > 0 || kk >= 3)) {
                        if (lastc && with_res && has_res) x8_wait_vm<allow + X8_NBIAS + X8_NRES>();
                        else if (lastc && with_res) x8_wait_vm<allow + X8_NBIAS>();
                        else x8_wait_vm<allow>();
                    }
                };
                // (MODE 2: the wait only has to precede the row's barrier, not this tap's MFMAs; placed behind them -- ADAS_H8X_WAIT_LAST builds --
                // it measured 0.5 % slower end to end, profiles/r06/ab_h8x3_variants.txt: the round-4 order stays)
#ifndef ADAS_H8X_WAIT_LAST
                counted_wait();
#else
                if (MODE == 1) counted_wait();
#endif
                if (MODE == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_setprio(1);
                if (lastc) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) xoff[j][kk] = tap_offset(apn[j], kk);
                    if constexpr (kk < NWP) gnxt[kk] = win_offset(nxt, kk);
                }
                if (wave_on) {
#pragma unroll
                    for (int i = islo ? 2 : 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = Fp16::mfma(wf[i], xf[j], acc[i][j]);
                }
                if (lastc) {
#pragma unroll
                    for (int g = 0; g < (islo ? 8 : 16); ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);   // address arithmetic of the next item in its shadow
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                if (sync_here) {
                    __builtin_amdgcn_sched_barrier(0);
#ifdef ADAS_H8X_WAIT_LAST
                    if (MODE != 1) counted_wait();
#endif
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); tap(std::integral_constant<int, 2>{});
            tap(std::integral_constant<int, 3>{}); tap(std::integral_constant<int, 4>{}); tap(std::integral_constant<int, 5>{});
            tap(std::integral_constant<int, 6>{}); tap(std::integral_constant<int, 7>{}); tap(std::integral_constant<int, 8>{});
#ifdef ADAS_H8X_PROF
            if (lastc) { H8XP(3) } else if (islo) { H8XP(2) } else { H8XP(1) }
#endif
        };
        for (int c = 0; c + 1 < a.nck; ++c) {
            if (c & 1) chunk(std::false_type{}, std::true_type{}, c);
            else chunk(std::false_type{}, std::false_type{}, c);
        }
        chunk(std::true_type{}, std::true_type{}, a.nck - 1);   // nck is even: the last half-chunk is an L chunk

        // ---------------- epilogue: lane holds channels kg*4..+3 of pixel lrow of every (i, j) tile
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        H8XP(4)
#if !defined(ADAS_H8X_SCALAR_EPI) && !defined(ADAS_H8X_NARROW_ST)
        auto write_out = [&](auto rm_c) {
            constexpr int RM = decltype(rm_c)::value;
            yf32x2 val[4][2][2];   // [pixel tile][channel tile][value pair]: the activated outputs
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    yf32x2 v0 = yf32x2{acc[i + 2][j][0], acc[i + 2][j][1]} * kX3Down + yf32x2{acc[i][j][0], acc[i][j][1]};
                    yf32x2 v1 = yf32x2{acc[i + 2][j][2], acc[i + 2][j][3]} * kX3Down + yf32x2{acc[i][j][2], acc[i][j][3]};
                    if (RM != RES_NONE) {
                        // the swap that forms the 16-byte pieces is its own inverse: it hands each lane its own hi and lo words back
                        const auto q0 = __builtin_amdgcn_permlane16_swap(rraw[j][i].x, rraw[j][i].z, false, false);
                        const auto q1 = __builtin_amdgcn_permlane16_swap(rraw[j][i].y, rraw[j][i].w, false, false);
                        const yf32x2 r0 = x8_join2(q0[0], q0[1]), r1 = x8_join2(q1[0], q1[1]);
                        if (RM == RES_BEFORE_ACT) { v0 = x8_act2<ACT>(v0 + r0); v1 = x8_act2<ACT>(v1 + r1); }
                        else { v0 = x8_act2<ACT>(v0) + r0; v1 = x8_act2<ACT>(v1) + r1; }
                    } else {
                        v0 = x8_act2<ACT>(v0); v1 = x8_act2<ACT>(v1);
                    }
                    val[j][i][0] = v0; val[j][i][1] = v1;
                    acc[i][j] = yf32x4{biasn[i].x, biasn[i].y, biasn[i].z, biasn[i].w};
                    acc[i + 2][j] = yf32x4{0.f, 0.f, 0.f, 0.f};
                }
            uint32_t hw[4][2][2];
            x8_hi_mode(true);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    hw[j][i][0] = x8_cvt_hi2(val[j][i][0]);
                    hw[j][i][1] = x8_cvt_hi2(val[j][i][1]);
                }
            x8_hi_mode(false);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t oo = po[j] == X8_OOB ? X8_OOB : (po[j] * (uint32_t)a.out_cs + (uint32_t)a.out_coff) * 4u + ch0;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    // channels past cout (the padded tail of the last 64-channel block) are computed on zero weights and not stored
                    const uint32_t oi = (cb * 64 + hb * 32 + i * 16 + (kg >> 1) * 8 < a.cout) ? oo + i * 64 : X8_OOB;   // (cout is a multiple of 8)
                    const uint32_t l0 = x8_lo2(val[j][i][0], hw[j][i][0]), l1 = x8_lo2(val[j][i][1], hw[j][i][1]);
                    // even rows end up with {own hi, partner's hi} = the group's 16 hi bytes, odd rows with {partner's lo, own lo}
                    const auto s0 = __builtin_amdgcn_permlane16_swap(hw[j][i][0], l0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(hw[j][i][1], l1, false, false);
                    __builtin_amdgcn_raw_buffer_store_b128(yu32x4{s0[0], s1[0], s0[1], s1[1]}, rout, oi, 0, 0);
                }
            }
        };
#else
        auto write_out = [&](auto rm_c) {
            constexpr int RM = decltype(rm_c)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t oo = po[j] == X8_OOB ? X8_OOB : (po[j] * (uint32_t)a.out_cs + (uint32_t)a.out_coff) * 4u + ch0;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    // channels past cout (the padded tail of the last 64-channel block) are computed on zero weights and not stored
#ifdef ADAS_H8X_NARROW_ST
                    const uint32_t oi = (cb * 64 + hb * 32 + i * 16 + kg * 4 < a.cout) ? oo + i * 64 : X8_OOB;
#else
                    const uint32_t oi = (cb * 64 + hb * 32 + i * 16 + (kg >> 1) * 8 < a.cout) ? oo + i * 64 : X8_OOB;   // (cout is a multiple of 8)
#endif
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + acc[i + 2][j][e] * kX3Down;
                    if (RM != RES_NONE) {
                        // (words copied to scalars first: hipcc's __builtin_bit_cast applied directly to a vector ELEMENT reads element 0)
#ifdef ADAS_H8X_NARROW_ST
                        const uint32_t wh0 = rraw[j][i].x, wh1 = rraw[j][i].y, wl0 = rraw[j][i].z, wl1 = rraw[j][i].w;
#else
                        // the swap that forms the 16-byte pieces is its own inverse: it hands each lane its own hi and lo words back
                        const auto q0 = __builtin_amdgcn_permlane16_swap(rraw[j][i].x, rraw[j][i].z, false, false);
                        const auto q1 = __builtin_amdgcn_permlane16_swap(rraw[j][i].y, rraw[j][i].w, false, false);
                        const uint32_t wh0 = q0[0], wl0 = q0[1], wh1 = q1[0], wl1 = q1[1];
#endif
                        const e_f16x2 h0 = __builtin_bit_cast(e_f16x2, wh0);
                        const e_f16x2 h1 = __builtin_bit_cast(e_f16x2, wh1);
                        const e_f16x2 l0 = __builtin_bit_cast(e_f16x2, wl0);
                        const e_f16x2 l1 = __builtin_bit_cast(e_f16x2, wl1);
                        const float r[4] = {x3_join(h0[0], l0[0]), x3_join(h0[1], l0[1]), x3_join(h1[0], l1[0]), x3_join(h1[1], l1[1])};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = RM == RES_BEFORE_ACT ? x8_act<ACT>(v[e] + r[e]) : x8_act<ACT>(v[e]) + r[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = x8_act<ACT>(v[e]);
                    }
                    e_f16x2 h0, h1, l0, l1;
                    _Float16 h, l;
                    x3_split(v[0], h, l); h0[0] = h; l0[0] = l;
                    x3_split(v[1], h, l); h0[1] = h; l0[1] = l;
                    x3_split(v[2], h, l); h1[0] = h; l1[0] = l;
                    x3_split(v[3], h, l); h1[1] = h; l1[1] = l;
#ifdef ADAS_H8X_NARROW_ST
                    __builtin_amdgcn_raw_buffer_store_b64(yu32x2{__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1)}, rout, oi, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(yu32x2{__builtin_bit_cast(uint32_t, l0), __builtin_bit_cast(uint32_t, l1)}, rout, oi + 16, 0, 0);
#else
                    {   // even rows end up with {own hi, partner's hi} = the group's 16 hi bytes, odd rows with {partner's lo, own lo}
                        const auto s0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, l0), false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, h1), __builtin_bit_cast(uint32_t, l1), false, false);
                        __builtin_amdgcn_raw_buffer_store_b128(yu32x4{s0[0], s1[0], s0[1], s1[1]}, rout, oi, 0, 0);
                    }
#endif
                    acc[i][j] = yf32x4{biasn[i].x, biasn[i].y, biasn[i].z, biasn[i].w};
                    acc[i + 2][j] = yf32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        };
#endif
        if (a.res_mode == RES_NONE) write_out(std::integral_constant<int, RES_NONE>{});
        else if (a.res_mode == RES_BEFORE_ACT) write_out(std::integral_constant<int, RES_BEFORE_ACT>{});
        else write_out(std::integral_constant<int, RES_AFTER_ACT>{});
        H8XP(5)
#ifdef ADAS_H8X_PROF
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
        for (int i = 0; i < NWP; ++i) gcur[i] = gnxt[i];
        wcur = wnxt_item;
        par = (par + a.nck) & 1;
    }
    if (MODE == 1 && !hb) __builtin_amdgcn_s_barrier();   // pairs with group 1's extra barrier
#ifdef ADAS_H8X_PROF
    if (lane == 0 && grp == 0) {
        unsigned long long* b__ = g_h8x_prof[blockIdx.x & 255];
        for (int i__ = 0; i__ < 16; ++i__)
            if (i__ != 7) atomicAdd(&b__[i__ + 16 * hb], pacc__[i__]);
        atomicAdd(&b__[7 + 16 * hb], (unsigned long long)nit__);
    }
#endif
}

#ifdef ADAS_H8X_PROF
extern "C" int adas_debug_h8x_prof(unsigned long long* out32, int reset) {
    static unsigned long long h[256][32];
    if (out32) {   // 32 values: 16 per wave group, summed over workgroups
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_h8x_prof), sizeof(h)) != hipSuccess) return -1;
        for (int i = 0; i < 32; ++i) {
            out32[i] = 0;
            for (int b = 0; b < 256; ++b) out32[i] += h[b][i];
        }
    }
    if (reset) {
        memset(h, 0, sizeof(h));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_h8x_prof), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// -------------------------------------------------------------------------------------
static int x8_mode() {   // ADAS_HALO8_X3: 0 off, 1 on (default: synchronisation variant picked per layer), 2 / 3 force variant 2 / 1
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_HALO8_X3");
        v = e ? atoi(e) : 1;
    }
    return v;
}

static int x8_policy() {   // ADAS_H8X_PLAN: 1 (default) = smallest window among the most efficient strip widths, 0 = the widest strip (rounds 4-5)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_H8X_PLAN");
        v = e ? atoi(e) : 1;
    }
    return v;
}
static bool x8_plan(int Ho, int Wo, HaloPlan* pl) { return plan_halo(Ho, Wo, 1, pl, X8_MAXPIX, 0, x8_policy()); }

// least share of the workgroup slots of the launch's last round that must be busy (ADAS_H8X_MIN_FILL, default 0.6; it was 0.8 until the
// end of round 6: YOLOv8l at 8 frames has 100 items per XCD = 0.78 and stayed on the generic kernel at 140-200 TFLOP/s)
static double x8_min_fill() {
    static double v = -1.0;
    if (v < 0) { const char* e = getenv("ADAS_H8X_MIN_FILL"); v = e ? atof(e) : 0.6; if (!(v > 0.0 && v <= 1.0)) v = 0.6; }
    return v;
}
static int x8_blocks_per_unit(long tiles8, int ncb) {
    for (int cpw = ncb; cpw >= 1; --cpw) {
        if (ncb % cpw) continue;
        const long units8 = tiles8 * (ncb / cpw), rounds = (units8 + X8_SLOTS - 1) / X8_SLOTS;
        if (units8 >= X8_SLOTS && (double)units8 / (double)(rounds * X8_SLOTS) >= x8_min_fill()) return cpw;
    }
    // a layer that gives every XCD 12-31 items (96-248 of the 256 CUs busy for one round: the 20x20 Detect convs at 64 frames) still
    // finishes sooner here than on the generic kernel (conv_x3_igemm: ~90 TFLOP/s on those shapes)
    if (tiles8 * ncb >= 12 && tiles8 * ncb < X8_SLOTS) return 1;
    return 0;
}

// static part (shapes): decides at load time whether a conv also gets the half-chunk weight packing.  Channel counts need not be whole
// blocks: Cout is padded to 64-channel items (zero weight rows, stores masked) and Cin to 32-channel chunks (zero weight columns; the
// window then reads past the view's channels into the neighbouring finite values, which the zero weights cancel).  The padding is MFMA
// work: layers that would spend more than half of it on zeros stay on the generic kernel.
static int x8_cin_pad(int cin) { return (cin + 31) / 32 * 32; }
static int x8_cout_pad(int cout) { return (cout + 63) / 64 * 64; }
bool halo8_x3_shape_ok(int kh, int kw, int stride, int pad, const TView& in, const TView& out) {
    if (!x8_mode()) return false;
    if (kh != 3 || kw != 3 || stride != 1 || pad != 1) return false;
    if (in.f32 || out.f32 || out.h != in.h || out.w != in.w) return false;
    if ((out.c & 7) || (in.c & 7) || in.c < 16) return false;
    if ((in.cs & 7) || (in.coff & 7) || (out.cs & 7) || (out.coff & 7)) return false;
    if ((double)in.c * out.c < 0.5 * (double)x8_cin_pad(in.c) * x8_cout_pad(out.c)) return false;
    HaloPlan pl;
    return x8_plan(out.h, out.w, &pl) && pl.eff >= 0.6 && pl.maxpix <= X8_MAXPIX;
}
size_t halo8_x3_weight_bytes(int cout, int cin) { return (size_t)(x8_cout_pad(cout) / 32) * (size_t)(2 * (x8_cin_pad(cin) / 32)) * X8_SLAB; }

bool halo8_x3_applicable(int kh, int kw, int stride, int pad, int n, const TView& in, const TView& out, const TView& res, int res_mode) {
    if (!halo8_x3_shape_ok(kh, kw, stride, pad, in, out)) return false;
    if (res_mode != RES_NONE && ((res.cs & 7) || (res.coff & 7) || res.f32)) return false;
    if ((double)n * in.h * in.w * in.cs * 4.0 >= (double)X8_OOB || (double)n * out.h * out.w * out.cs * 4.0 >= (double)X8_OOB) return false;
    if (res_mode != RES_NONE && (double)n * out.h * out.w * res.cs * 4.0 >= (double)X8_OOB) return false;
    if ((double)halo8_x3_weight_bytes(out.c, in.c) >= (double)X8_OOB) return false;
    HaloPlan pl;
    if (!x8_plan(out.h, out.w, &pl)) return false;
    const long ntiles = (long)n * pl.NS * pl.TPS, tiles8 = (ntiles + 7) / 8;
    if (ntiles * (x8_cout_pad(out.c) / 64) * pl.NS * pl.TPS >= (1L << 32)) return false;
    return x8_blocks_per_unit(tiles8, x8_cout_pad(out.c) / 64) > 0;
}

// fp32 [cout][9][cin] -> halves [cout_pad / 32][2 * cin_pad / 32][9][64 rows][32]: block b = output channels 32 b .. 32 b + 31;
// rows 0-31 main, rows 32-63 cross.  H half-chunk (even): main = hi(w), cross = lo(w); L half-chunk (odd): main = 0, cross = hi(w).
// Rows past cout and columns past cin are zero.
__global__ void pack_weights_h8x3_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int cout, int cin, int nck, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i & 31);
        size_t r = i >> 5;
        const int row = (int)(r & 63); r >>= 6;
        const int tap = (int)(r % 9); r /= 9;
        const int ck = (int)(r % nck);
        const size_t blk = r / nck;
        const size_t co = blk * 32 + (row & 31);
        const int ci = (ck >> 1) * 32 + k;
        const float w = (co < (size_t)cout && ci < cin) ? src[(co * 9 + tap) * cin + ci] : 0.0f;
        _Float16 h, l;
        x3_split(w, h, l);
        const bool cross = row >= 32, lo_chunk = (ck & 1) != 0;
        const _Float16 v = lo_chunk ? (cross ? h : (_Float16)0.0f) : (cross ? l : h);
        dst[i] = __builtin_bit_cast(uint16_t, v);
    }
}
hipError_t launch_pack_weights_h8x3(const float* src, void* dst, int cout, int cin, hipStream_t st) {
    if ((cout & 7) || (cin & 7)) return hipErrorInvalidValue;
    const size_t total = halo8_x3_weight_bytes(cout, cin) / 2;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_weights_h8x3_kernel, dim3(blocks), dim3(256), 0, st, src, (uint16_t*)dst, cout, cin, 2 * (x8_cin_pad(cin) / 32), total);
    return hipGetLastError();
}

static bool x8_lhalf() {   // ADAS_H8X_LHALF=0: fetch whole weight tiles for the L half-chunks too (the round-4 / 5 kernel)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_H8X_LHALF");
        v = e ? atoi(e) : 1;
    }
    return v != 0;
}

static bool x8_share() {   // ADAS_H8X_SHARE=0: every half-chunk fetches its own weight tiles (the round-4 / 5 stream)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_H8X_SHARE");
        v = e ? atoi(e) : 1;
    }
    return v != 0;
}

template <int NWP>
static hipError_t x8_launch_sh(const H8XDev& d, int act, dim3 grid, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_h8x3_kernel<ACT_NONE, 2, false, true, NWP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_h8x3_kernel<ACT_SILU, 2, false, true, NWP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_h8x3_kernel<ACT_RELU, 2, false, true, NWP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_h8x3_kernel<ACT_LEAKY, 2, false, true, NWP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    if (act == ACT_SILU) hipLaunchKernelGGL((conv_h8x3_kernel<ACT_SILU, 2, false, true, NWP>), grid, dim3(X8_THR), X8_LDS, st, d);
    else if (act == ACT_RELU) hipLaunchKernelGGL((conv_h8x3_kernel<ACT_RELU, 2, false, true, NWP>), grid, dim3(X8_THR), X8_LDS, st, d);
    else if (act == ACT_LEAKY) hipLaunchKernelGGL((conv_h8x3_kernel<ACT_LEAKY, 2, false, true, NWP>), grid, dim3(X8_THR), X8_LDS, st, d);
    else hipLaunchKernelGGL((conv_h8x3_kernel<ACT_NONE, 2, false, true, NWP>), grid, dim3(X8_THR), X8_LDS, st, d);
    return hipGetLastError();
}

template <int MODE, bool LH>
static hipError_t x8_launch(const H8XDev& d, int act, dim3 grid, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_h8x3_kernel<ACT_NONE, MODE, LH, false, X8_NWP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_h8x3_kernel<ACT_SILU, MODE, LH, false, X8_NWP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_h8x3_kernel<ACT_RELU, MODE, LH, false, X8_NWP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_h8x3_kernel<ACT_LEAKY, MODE, LH, false, X8_NWP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    if (act == ACT_SILU) hipLaunchKernelGGL((conv_h8x3_kernel<ACT_SILU, MODE, LH, false, X8_NWP>), grid, dim3(X8_THR), X8_LDS, st, d);
    else if (act == ACT_RELU) hipLaunchKernelGGL((conv_h8x3_kernel<ACT_RELU, MODE, LH, false, X8_NWP>), grid, dim3(X8_THR), X8_LDS, st, d);
    else if (act == ACT_LEAKY) hipLaunchKernelGGL((conv_h8x3_kernel<ACT_LEAKY, MODE, LH, false, X8_NWP>), grid, dim3(X8_THR), X8_LDS, st, d);
    else hipLaunchKernelGGL((conv_h8x3_kernel<ACT_NONE, MODE, LH, false, X8_NWP>), grid, dim3(X8_THR), X8_LDS, st, d);
    return hipGetLastError();
}

hipError_t launch_conv_halo8_x3(const ConvArgs& a, hipStream_t st) {
    HaloPlan pl;
    if (!a.wgt_h8x3 || !halo8_x3_applicable(a.kh, a.kw, a.stride, a.pad, a.n, a.in, a.out, a.res, a.res_mode) || !x8_plan(a.out.h, a.out.w, &pl))
        return hipErrorNotSupported;
    {   // ADAS_H8X_SW=<strip width>: narrower strips = squarer tiles = smaller windows (less halo re-read) at more padded pixels; the
        // tile grid is then re-checked against the item-count rule below (an experiment knob, like conv_halo8's ADAS_H8_SW)
        static int sw = -1;
        if (sw < 0) { const char* e = getenv("ADAS_H8X_SW"); sw = e ? atoi(e) : 0; }
        HaloPlan alt;
        if (sw > 0 && plan_halo_sw(a.out.h, a.out.w, 1, sw, X8_MAXPIX, &alt) && alt.eff >= 0.6) {
            const long nt = (long)a.n * alt.NS * alt.TPS;
            if (x8_blocks_per_unit((nt + 7) / 8, x8_cout_pad(a.out.c) / 64) > 0) pl = alt;
        }
    }
    H8XDev d;
    d.in = a.in.p; d.wgt = a.wgt_h8x3; d.bias = a.bias; d.out = a.out.p; d.res = a.res.p;
    d.in_cs = a.in.cs; d.in_coff = a.in.coff; d.cin = a.in.c; d.H = a.in.h; d.W = a.in.w;
    d.out_cs = a.out.cs; d.out_coff = a.out.coff; d.cout = a.out.c;
    d.res_cs = a.res.cs; d.res_coff = a.res.coff; d.res_mode = a.res_mode;
    d.nck = 2 * (x8_cin_pad(a.in.c) / 32);
    d.in_bytes = (uint32_t)((size_t)a.n * a.in.h * a.in.w * a.in.cs * 4);
    d.wgt_bytes = (uint32_t)halo8_x3_weight_bytes(a.out.c, a.in.c);
    d.out_bytes = (uint32_t)((size_t)a.n * a.out.h * a.out.w * a.out.cs * 4);
    d.res_bytes = a.res_mode != RES_NONE ? (uint32_t)((size_t)a.n * a.out.h * a.out.w * a.res.cs * 4) : 0u;
    d.SW = pl.SW; d.NS = pl.NS; d.TPS = pl.TPS; d.WW = pl.WW;
    d.mg_ww = pl.mg_ww; d.mg_sw = pl.mg_sw;
    d.mg_img = (uint32_t)(((1ull << 32) + (uint64_t)(pl.NS * pl.TPS) - 1) / (uint64_t)(pl.NS * pl.TPS));
    d.mg_tps = (uint32_t)(((1ull << 32) + (uint64_t)pl.TPS - 1) / (uint64_t)pl.TPS);
    d.ntiles = a.n * pl.NS * pl.TPS;
    d.tiles8 = (d.ntiles + 7) / 8;
    d.ncb = x8_cout_pad(a.out.c) / 64;
    d.cpw = x8_blocks_per_unit(d.tiles8, d.ncb);
    if (d.cpw <= 0) return hipErrorNotSupported;
    const int upt = d.ncb / d.cpw;
    d.mg_upt = (uint32_t)(((1ull << 32) + (uint64_t)upt - 1) / (uint64_t)upt);
    const int units8 = d.tiles8 * upt;
    // workgroups per XCD over the launch (ADAS_H8X_SLOTS, default 32 = one per CU, every one persistent over its share of the items; more:
    // the later workgroups are dealt as CUs fall free, each with fewer items)
    static int max_slots = -1;
    if (max_slots < 0) { const char* e = getenv("ADAS_H8X_SLOTS"); max_slots = e ? atoi(e) : X8_SLOTS; if (max_slots < 8 || max_slots > 4096) max_slots = X8_SLOTS; }
    const int slots = units8 < max_slots ? units8 : max_slots;
    dim3 grid(8 * slots);
    const int forced = x8_mode();
    // (the wave-groups-a-barrier-apart variant, MODE 1, lost to MODE 2 on every layer once both were measured at one commit --
    // profiles/r06/ab_h8x3.txt: -4.3 % end to end when forced everywhere, +-0.3 % on the nck >= 24 layers it used to take; ADAS_HALO8_X3=3 keeps it reachable)
    const bool pingpong = forced == 3;
    if (x8_share() && forced != 3) {   // shared weight tiles (one barrier per tap row), as many window pieces as this plan's windows need
        const int nwp = (pl.maxpix + 127) / 128;
        if (nwp <= 3) return x8_launch_sh<3>(d, a.act, grid, st);
        if (nwp == 4) return x8_launch_sh<4>(d, a.act, grid, st);
        return x8_launch_sh<5>(d, a.act, grid, st);
    }
    if (x8_lhalf()) return pingpong ? x8_launch<1, true>(d, a.act, grid, st) : x8_launch<2, true>(d, a.act, grid, st);
    return pingpong ? x8_launch<1, false>(d, a.act, grid, st) : x8_launch<2, false>(d, a.act, grid, st);
}

}  // namespace adas
