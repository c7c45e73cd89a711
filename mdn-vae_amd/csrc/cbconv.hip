// The conv/ResNet encoder of function/VAE_CNN_NN.py:83-98,110-134 (and its copies in function/VAE-CNN-GCN.py,
// function_FC/VAE_CNN_GCN.py) on 16-bit operands that never pass through a register on their way to the matrix
// cores.
//
// Activations that feed a 3x3 convolution live in a channel-blocked layout
//     act[part][group-plane][sample][position][8 channels]            (16-bit elements, 16 bytes per position)
// written once by the pass that produces them (instance norm + ReLU forward, instance-norm backward): already
// padded (reflected border, zero border, or the four parity planes of a stride-2 layer's input), already rounded
// (part 0 = bf16) or split (two fp16 parts of the value times a power of two).  A position's 16 bytes are exactly
// one lane's share of an MFMA A operand (8 consecutive k = 8 channels), positions of one group-plane are contiguous
// across the whole batch, and a 3x3 tap is a constant offset in positions.  So a convolution is a GEMM whose A tile
// is a slab of positions copied by LDS-DMA (global_load_lds, 16 bytes per lane) and read back at nine offsets, and
// whose B tile is a stream of pre-packed weight fragments, also LDS-DMA.  Outputs leave as fp32 planes
//     out[sample][channel][row * pitch + column]
// over the same pitched grid as the input positions (columns >= the plane's width are scratch).
//
// Stride-2 layers (function/VAE_CNN_NN.py:118,124) read four parity planes (space-to-depth): tap (ky, kx) is plane
// ((ky+1) & 1, (kx+1) & 1) at block offset (ky > 0, kx > 0), so they run on the stride-1 kernel.  Their data gradient
// is a 2x2-tap stride-1 convolution of the output gradient that writes the four parity planes of the input gradient.
#include <type_traits>
#include "common.h"
#include "../../include/mdnvae_hip.h"

namespace mdnvae {

#define CB_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define CB_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

typedef __bf16 cb_bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 cb_h8 __attribute__((ext_vector_type(8)));
typedef float cb_acc16 __attribute__((ext_vector_type(16)));
typedef float cb_f4 __attribute__((ext_vector_type(4)));
typedef float cb_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float cb_f2 __attribute__((ext_vector_type(2)));
typedef float cb_f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef __bf16 cb_h4u __attribute__((ext_vector_type(4), aligned(2)));
typedef unsigned cb_u4 __attribute__((ext_vector_type(4)));
typedef unsigned cb_u2 __attribute__((ext_vector_type(2)));

constexpr int CB_MAXTAPS = 9;
constexpr int CB_NBUF = 4;            // weight stage buffers: three stages in flight
constexpr int CB_ROWS = 256;          // output rows (positions) per workgroup

// Largest-magnitude records (MDNVAE_ABSMAX_SLOTS partial maxima, bit patterns of floats) give the power-of-two scales of
// tensors stored as two fp16 parts: the scale brings the recorded magnitude into [2^14, 2^15).
__device__ __forceinline__ float cb_pow2_scale(unsigned bits) {
    const int e = (int)((bits >> 23) & 0xffu);
    if (e == 0 || e == 255) return 1.0f;
    const int se = 268 - e;
    return __uint_as_float((unsigned)(se < 253 ? se : 253) << 23);
}
static_assert(MDNVAE_ABSMAX_SLOTS == 64, "one slot per lane");
__device__ __forceinline__ unsigned cb_slot_max(const unsigned* __restrict__ slots) {      // whole waves only
    unsigned m = slots[threadIdx.x & 63];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    return m;
}
// record of a wave's largest magnitude, spread over the slots: an atomic only when the slot (read past the L1, which
// may hold an older, smaller value) does not hold as much already -- tens of thousands of waves share 64 words, and
// after the first few almost none of them has anything to add
__device__ __forceinline__ void cb_record_max(unsigned* __restrict__ slots, unsigned m, unsigned spread) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) {
        unsigned* slot = slots + (spread & (MDNVAE_ABSMAX_SLOTS - 1));
        if (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < m) atomicMax(slot, m);
    }
}

struct CbConv {
    const char* act;        // channel-blocked input, part 0
    long part_stride;       // bytes between parts
    const char* w;          // packed weight fragments [N block][chunk][stage][tap][n tile][part][1 KiB]
    const float* bias;      // [ntot] or null
    float* out;             // [B][ntot][Qr]
    int B, PP;              // samples; positions per (group-plane, sample) of the input
    int nchunk;             // 16-channel chunks of the reduction
    int Qr;                 // output rows per sample (multiple of 4)
    int halo;               // largest tap offset in positions
    int ntot;               // output planes per sample
    int tap_lds[CB_MAXTAPS + 3];   // byte offset of tap t in the slab: (plane * 2 * slab positions + offset) * 16
    float out_scale;        // accumulators are multiplied by this (undoes a fixed power-of-two scale of fp16 parts)
    const unsigned* amax_a; // records the input / the weights were scaled by (their scales are undone too), or null
    const unsigned* amax_w;
};

template <int N>
__device__ __forceinline__ void cb_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// PARTS: 1 = bf16 operands, 2 = two fp16 parts per operand (hh + hl + lh, 2^-22 per product).
// NGP: group-planes per 16-channel chunk (2: k halves; 8: four parity planes x k halves).  NJ: 64-position blocks per
// group-plane in the slab.  WGM x WGN waves of WM x WN 32x32 tiles.  TPS taps per stage, SPC stages per chunk.
//
// Pipeline.  Stage s = (chunk c, r) is TPS taps of one 16-channel chunk.  Weights of stage s + 4 and (at the end of
// a chunk) the slab of chunk c + 2 are issued at the gate of stage s + 1, which stands in front of stage s's last
// tap: at that point this wave's own pieces of stage s + 1 have landed (a counted vmcnt that leaves the two younger
// weight stages, and a younger slab, in flight) and every LDS read of stage s is complete; the barrier makes both
// true for all waves.  Fragments are read one tap ahead of the MFMAs that use them, across stage boundaries too.
// Past the end the last stage / chunk is fetched again into buffers nobody reads any more, which keeps the counts
// the same in every iteration.  NSB = 1 (one slab buffer, where two do not fit): the next chunk's slab is requested
// after the chunk's last gate and waited for on the spot.
//
// PERSIST: one workgroup per CU walks tiles blockIdx.x, + gridDim.x, ...: the first operands of the next tile are
// requested before this tile's epilogue, and the epilogue's stores drain to memory under the next tile's MFMAs
// (with one workgroup per CU and a grid of tiles, the matrix cores sat idle for the whole write-out: 28 % of the
// kernel).  The epilogue then has an LDS region of its own behind the stage buffers.
template <int PARTS, int NGP, int NJ, int WGM, int WGN, int WM, int WN, int TPS, int SPC, bool PERSIST, int NSW = 8, int NSB = 2, int OUT16 = 0>
__global__ __launch_bounds__(512) void cb_conv_kernel(CbConv p, int nblk, int ntiles) {
    static_assert(WGM * WGN == 8 && WGM * WM * 32 == CB_ROWS, "eight waves cover 256 rows");
    constexpr int NT = WGN * WN;                              // 32-channel tiles per workgroup
    constexpr int WSTAGE = TPS * NT * PARTS * 1024;           // bytes of one weight stage
    constexpr int WPIECES = TPS * NT * PARTS;                 // 1 KiB pieces per weight stage
    // NSW waves (0 .. NSW - 1) issue the LDS-DMA; with NSW = 4 the other four write the results out, so that no wave
    // ever waits for its own stores when it waits for operands (one counter covers a wave's loads and stores, in order)
    static_assert(NSW == 8 || NSW == 4, "all waves stage, or four stage and four store");
    constexpr int WQ = WPIECES / NSW, WREM = WPIECES % NSW;
    constexpr int SLABPOS = NJ * 64;
    constexpr int SLABPART = NGP * SLABPOS * 16;              // bytes of one part of a slab
    constexpr int SLAB = PARTS * SLABPART;
    constexpr int SPIECES = PARTS * NGP * NJ;
    static_assert(SPIECES % NSW == 0, "every staging wave stages the same number of slab pieces");
    constexpr int SP = SPIECES / NSW;
    constexpr int MAIN = CB_NBUF * WSTAGE + NSB * SLAB;       // NSB slab buffers: chunks c .. c + NSB - 1 resident or in flight
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* wring = smem;                                       // [CB_NBUF][WSTAGE]
    char* slabs = smem + CB_NBUF * WSTAGE;                    // [NSB][SLAB]
    float* ep = reinterpret_cast<float*>(smem + (PERSIST ? MAIN : 0));     // epilogue transpose buffer

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid / WGN, wn = wid - wm * WGN;
    const int h = lane >> 5, l32 = lane & 31;
    const int nstage = p.nchunk * SPC;
    const long chunk_stride = (long)NGP * p.B * p.PP * 16;
    const long Rend = (long)p.B * p.Qr;
    float oscale = p.out_scale;
    if (p.amax_a) oscale /= cb_pow2_scale(cb_slot_max(p.amax_a));
    if (p.amax_w) oscale /= cb_pow2_scale(cb_slot_max(p.amax_w));

    int taps[TPS * SPC];
#pragma unroll
    for (int t = 0; t < TPS * SPC; ++t) taps[t] = p.tap_lds[t];

    // ---- per-tile state ----
    struct Tile {
        long R0;
        int nb, s0, qa, n0;
    };
    auto tile_of = [&](int t) {
        Tile T;
        const int rt = t / nblk;
        T.nb = t - rt * nblk;
        T.R0 = (long)rt * CB_ROWS;
        T.s0 = (int)(T.R0 / p.Qr);
        T.qa = (int)(T.R0 - (long)T.s0 * p.Qr);
        T.n0 = min(CB_ROWS, p.Qr - T.qa);                     // rows of sample s0; the rest belong to s0 + 1
        return T;
    };
    // staging plan of this wave for a tile: slab piece u = wid + NSW i of a chunk is (64-position block j, part,
    // group-plane), the group-plane running fastest
    long ssrc[SP];
    int sdst[SP];
    int lane_addr[WM];
    const char* wsrc;
    auto plan = [&](const Tile& T) {
        const int L0 = T.n0 + p.halo;                         // slab positions of the first segment
        const int s1off = (T.s0 + 1 < p.B) ? p.PP : 0;        // where the second segment's positions come from
#pragma unroll
        for (int i = 0; i < SP; ++i) {
            const int u = (wid & (NSW - 1)) + NSW * i;
            const int j = u / (PARTS * NGP), pg = u - j * (PARTS * NGP);
            const int part = pg / NGP, gp = pg - part * NGP;
            const int pos = j * 64 + lane;
            int src = pos < L0 ? T.qa + pos : s1off + (pos - L0);
            src = min(src, 2 * p.PP - 1);
            ssrc[i] = part * p.part_stride + (((long)gp * p.B + T.s0) * p.PP + src) * 16;
            sdst[i] = part * SLABPART + (gp * SLABPOS + j * 64) * 16;
        }
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const int i = (wm * WM + m) * 32 + l32;
            lane_addr[m] = (h * SLABPOS + (i < T.n0 ? i : L0 + (i - T.n0))) * 16;
        }
        wsrc = p.w + (long)T.nb * nstage * WSTAGE + lane * 16;
    };
    auto stage_slab = [&](int c, int buf) {
        if (NSW < 8 && wid >= NSW) return;
#pragma unroll
        for (int i = 0; i < SP; ++i)
            __builtin_amdgcn_global_load_lds(CB_GLOBAL_PTR(p.act + ssrc[i] + c * chunk_stride),
                                             CB_LDS_PTR(slabs + buf * SLAB + sdst[i]), 16, 0, 0);
    };
    auto stage_w = [&](int s, int buf) {
        if (NSW < 8 && wid >= NSW) return;
#pragma unroll
        for (int i = 0; i < WQ + (WREM ? 1 : 0); ++i) {
            if (i < WQ || wid < WREM) {
                const int q = wid + NSW * i;
                __builtin_amdgcn_global_load_lds(CB_GLOBAL_PTR(wsrc + (long)s * WSTAGE + q * 1024),
                                                 CB_LDS_PTR(wring + buf * WSTAGE + q * 1024), 16, 0, 0);
            }
        }
    };
    // first requests of a tile: slab of chunk 0, weights of stages 0..2
    auto prologue = [&]() {
        stage_slab(0, 0);
        if (NSB == 3) stage_slab(min(1, p.nchunk - 1), 1);
#pragma unroll
        for (int s = 0; s < CB_NBUF - 1; ++s) stage_w(min(s, nstage - 1), s);
    };

    auto mfma = [](const cb_bf8& a, const cb_bf8& b, const cb_acc16& c) {
        if constexpr (PARTS == 2)
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cb_h8, a), __builtin_bit_cast(cb_h8, b), c, 0, 0, 0);
        else
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    };
    cb_acc16 acc[WM][WN];
    // Fragments of one tap: A from the slab at the tap's offset, B from the stage's weight buffer.
    struct Frags {
        cb_bf8 a[PARTS][WM], b[PARTS][WN];
    };
    auto load_tap = [&](Frags& f, const char* wb, const char* sl, int tap_in_chunk, int tt) {
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) {
#pragma unroll
            for (int n = 0; n < WN; ++n)
                f.b[pt][n] = *reinterpret_cast<const cb_bf8*>(wb + ((tt * NT + wn * WN + n) * PARTS + pt) * 1024 + lane * 16);
#pragma unroll
            for (int m = 0; m < WM; ++m)
                f.a[pt][m] = *reinterpret_cast<const cb_bf8*>(sl + pt * SLABPART + lane_addr[m] + taps[tap_in_chunk]);
        }
    };
    auto mma_tap = [&](const Frags& f) {
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                if constexpr (PARTS == 2) {
                    acc[m][n] = mfma(f.a[0][m], f.b[1][n], acc[m][n]);
                    acc[m][n] = mfma(f.a[1][m], f.b[0][n], acc[m][n]);
                }
                acc[m][n] = mfma(f.a[0][m], f.b[0][n], acc[m][n]);
            }
    };
    // gate of a stage whose position in its chunk is r (see the head comment)
    auto gate = [&](int r) {
        const bool slab_young = NSB > 1 && (r == 1 || r == 2);
        if (NSW < 8 && wid >= NSW) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (WREM && wid < WREM) {
            if (slab_young) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (WQ + 1) + SP) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (WQ + 1)) : "memory");
        } else {
            if (slab_young) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * WQ + SP) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * WQ) : "memory");
        }
        __builtin_amdgcn_s_barrier();
    };

    // biases of this lane's output channels.  Loaded once where every tile has the same channels (one block of output
    // channels): an ordinary load inside the tile loop makes the compiler wait for ALL of the wave's memory operations
    // where its value is first used, the stores of the previous tile included.
    float bv[WN];
    auto load_bias = [&](int nb) {
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int ch = (nb * NT + wn * WN + n) * 32 + l32;
            bv[n] = (p.bias && ch < p.ntot) ? p.bias[ch] : 0.0f;
        }
#pragma unroll
        for (int n = 0; n < WN; ++n) asm volatile("" : "+v"(bv[n]));
    };
    int tile = blockIdx.x;
    Tile T = tile_of(tile);
    load_bias(T.nb);
    plan(T);
    prologue();
    while (true) {
        if (nblk > 1) load_bias(T.nb);
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;

        gate(0);
        if (NSB > 1) stage_slab(min(NSB - 1, p.nchunk - 1), NSB - 1);
        stage_w(min(CB_NBUF - 1, nstage - 1), CB_NBUF - 1);
        Frags cur, nxt;
        load_tap(cur, wring, slabs, 0, 0);
        int s = 0;
        for (int c = 0; c < p.nchunk; ++c) {
#pragma unroll
            for (int r = 0; r < SPC; ++r, ++s) {
                const char* wb = wring + (s % CB_NBUF) * WSTAGE;
                const char* sl = slabs + (c % NSB) * SLAB;
#pragma unroll
                for (int tt = 0; tt + 1 < TPS; ++tt) {
                    load_tap(nxt, wb, sl, r * TPS + tt + 1, tt + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_tap(cur);
                    __builtin_amdgcn_sched_barrier(0);
                    cur = nxt;
                }
                const int r1 = (r + 1) % SPC;
                gate(r1);
                if (r1 == 0) stage_slab(min(c + NSB, p.nchunk - 1), c % NSB);
                stage_w(min(s + CB_NBUF, nstage - 1), s % CB_NBUF);
                if (NSB == 1 && r1 == 0 && s + 1 < nstage) {
                    // one slab buffer (slabs too large for two): the next chunk is requested only now that nobody reads
                    // this one any more, and has to land before its first tap is read
                    cb_wait_vm<0>();
                    __builtin_amdgcn_s_barrier();
                }
                if (s + 1 < nstage)
                    load_tap(nxt, wring + ((s + 1) % CB_NBUF) * WSTAGE, slabs + ((r1 == 0 ? c + 1 : c) % NSB) * SLAB, r1 * TPS, 0);
                __builtin_amdgcn_sched_barrier(0);
                mma_tap(cur);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        }
        if (NSW == 8 || wid < NSW) cb_wait_vm<0>();            // the refetched last stages: nothing may land later
        __builtin_amdgcn_s_barrier();                          // every wave is done with the stage buffers and slabs
        // the next tile's first operands are requested now and arrive under this tile's epilogue
        const Tile Tc = T;
        const int next = tile + gridDim.x;
        const bool more = PERSIST && next < ntiles;
        if (more) {
            T = tile_of(next);
            plan(T);
            prologue();
        }
        // Epilogue.  Accumulator registers 4g..4g+3 of a 32 x 32 tile are rows 8g + 4h .. + 3 of column l32: a lane holds
        // 16 rows of ONE channel, and the output is channel-major planes, so the tiles cross through LDS, [channel][row],
        // and leave as runs of 128 consecutive rows of a channel (runs of 32 rows, which a pass per tile row would give,
        // cost a quarter of the kernel: every run ends in partially written lines).  One pass per (tile column n, half
        // of the rows): WGN * 32 channels x 128 rows, written by the waves that own those rows.
        {
            constexpr int ROWS_P = 128, ROWP = ROWS_P + 4;         // row pitch = 4 mod 32: a store group's 8 lanes hit different banks
            constexpr int RPL = OUT16 ? 8 : 4;                     // rows per lane in the read-back (16 bytes of output)
            constexpr int LPC = ROWS_P / RPL;                      // lanes per channel in the read-back
            constexpr int CPI = 64 / LPC;                          // channels per read-back instruction
            constexpr int NSTW = NSW == 8 ? 8 : 4;                 // waves that write out (the last NSTW)
            constexpr int NI = WGN * 32 / (NSTW * CPI);            // read-back instructions per storing wave and pass
            static_assert(NI >= 1 && NI * NSTW * CPI == WGN * 32, "the storing waves cover a pass exactly");
            constexpr int WROWS = WM * 32;                         // rows of a wave
            const int swid = wid - (8 - NSTW);                     // < 0: this wave does not store
            const int rp = (lane % LPC) * RPL;                     // row within the pass
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
                for (int hr = 0; hr < CB_ROWS / ROWS_P; ++hr) {
                    if ((wm * WROWS) / ROWS_P == hr) {
#pragma unroll
                        for (int m = 0; m < WM; ++m)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                cb_f4 o;
                                o.x = acc[m][n][4 * g] * oscale + bv[n];
                                o.y = acc[m][n][4 * g + 1] * oscale + bv[n];
                                o.z = acc[m][n][4 * g + 2] * oscale + bv[n];
                                o.w = acc[m][n][4 * g + 3] * oscale + bv[n];
                                *reinterpret_cast<cb_f4*>(ep + (wn * 32 + l32) * ROWP + (wm * WROWS) % ROWS_P + m * 32 + 8 * g + 4 * h) = o;
                            }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    const int i = hr * ROWS_P + rp;                // row of the 256-row tile
                    const int sg = i >= Tc.n0;
                    const int q = sg ? i - Tc.n0 : Tc.qa + i;
                    const bool row_ok = Tc.R0 + i < Rend;
#pragma unroll
                    for (int k = 0; k < NI; ++k) {
                        if (swid < 0) break;
                        const int chl = (swid * NI + k) * CPI + lane / LPC;
                        const int ch = (Tc.nb * NT + (chl >> 5) * WN + n) * 32 + (chl & 31);
                        const long at = ((long)(Tc.s0 + sg) * p.ntot + ch) * p.Qr + q;
                        const bool live = ch < p.ntot && row_ok;
                        const cb_f4 v = *reinterpret_cast<const cb_f4*>(ep + chl * ROWP + rp);
                        if constexpr (OUT16 == 2) {
                            // bf16 output (data gradients of the bf16 mode: what reads them rounds its own result to bf16)
                            const cb_f4 v2 = *reinterpret_cast<const cb_f4*>(ep + chl * ROWP + rp + 4);
                            cb_bf8 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                o[e] = (__bf16)v[e];
                                o[4 + e] = (__bf16)v2[e];
                            }
                            if (live) *reinterpret_cast<cb_bf8*>(reinterpret_cast<__bf16*>(p.out) + at) = o;
                        } else if constexpr (OUT16 == 1) {
                            // fp16 output: 11 significant bits, clamped to the finite range
                            const cb_f4 v2 = *reinterpret_cast<const cb_f4*>(ep + chl * ROWP + rp + 4);
                            cb_h8 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                o[e] = (_Float16)fminf(fmaxf(v[e], -65504.0f), 65504.0f);
                                o[4 + e] = (_Float16)fminf(fmaxf(v2[e], -65504.0f), 65504.0f);
                            }
                            if (live) *reinterpret_cast<cb_h8*>(reinterpret_cast<_Float16*>(p.out) + at) = o;
                        } else {
                            if (live) *reinterpret_cast<cb_f4*>(p.out + at) = v;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
        }
        if (!more) break;
        tile = next;
    }
}

// Weight fragments for cb_conv_kernel from w [Cout][Cin][3][3].
//   mode 0: forward (rows of the product = output channels, reduction over input channels, tap = ky * 3 + kx)
//   mode 1: data gradient of a stride-1 layer (rows = input channels, reduction over output channels, taps flipped)
//   mode 2: data gradient of a stride-2 / pad-1 layer as a 2x2-tap convolution: row = parity plane (py, px) * Cin + ci,
//           tap (ey, ex); kernel row ky = {py = 0: ey = 0 -> 1;  py = 1: ey = 0 -> 2, ey = 1 -> 0}, same for columns
// out [N block][chunk][stage][tap in stage][n tile][part][lane][8]: lane = (k half, row in the tile), entry j = reduction
// index 16 chunk + 8 half + j.  parts = 1: bf16; parts = 2: two fp16 parts of w * scale.
__device__ __forceinline__ void cb_pack_entry(const float* __restrict__ w, int Cout, int Cin, int mode, int nblk, int NT,
                                              int nchunk, int SPC, int TPS, int parts, const unsigned* __restrict__ amax,
                                              char* __restrict__ out, long idx) {
    const long total = (long)nblk * nchunk * SPC * TPS * NT * parts * 512;
    if (idx >= total) return;
    const int j = (int)(idx & 7), ln = (int)((idx >> 3) & 63);
    long r = idx >> 9;
    const int part = (int)(r % parts); r /= parts;
    const int n = (int)(r % NT); r /= NT;
    const int tt = (int)(r % TPS); r /= TPS;
    const int st = (int)(r % SPC); r /= SPC;
    const int c = (int)(r % nchunk);
    const int nb = (int)(r / nchunk);
    const int tap = st * TPS + tt;
    const int row = (nb * NT + n) * 32 + (ln & 31);
    const int k = c * 16 + 8 * (ln >> 5) + j;
    float v = 0.0f;
    if (mode == 0) {
        if (row < Cout && k < Cin) v = w[((long)row * Cin + k) * 9 + tap];
    } else if (mode == 1) {
        if (row < Cin && k < Cout) v = w[((long)k * Cin + row) * 9 + (8 - tap)];
    } else {
        const int plane = row / Cin, ci = row - plane * Cin;
        if (plane < 4 && k < Cout) {
            const int py = plane >> 1, px = plane & 1, ey = tap >> 1, ex = tap & 1;
            const int ky = py == 0 ? (ey == 0 ? 1 : -1) : (ey == 0 ? 2 : 0);
            const int kx = px == 0 ? (ex == 0 ? 1 : -1) : (ex == 0 ? 2 : 0);
            if (ky >= 0 && kx >= 0) v = w[((long)k * Cin + ci) * 9 + ky * 3 + kx];
        }
    }
    if (parts == 1) {
        reinterpret_cast<__bf16*>(out)[idx] = (__bf16)v;
    } else {
        // scale: the power of two that brings the tensor's largest magnitude into [2^14, 2^15)
        unsigned m = 0;
        for (int i = 0; i < MDNVAE_ABSMAX_SLOTS; ++i) m = max(m, amax[i]);
        const int e = (int)((m >> 23) & 0xffu);
        float sc = 1.0f;
        if (e != 0 && e != 255) {
            const int se = 268 - e;
            sc = __uint_as_float((unsigned)(se < 253 ? se : 253) << 23);
        }
        const float t = v * sc;
        const _Float16 hi = (_Float16)t;
        reinterpret_cast<_Float16*>(out)[idx] = part == 0 ? hi : (_Float16)(t - (float)hi);
    }
}

__global__ __launch_bounds__(256) void cb_pack_weights_kernel(const float* __restrict__ w, int Cout, int Cin, int mode,
                                                              int nblk, int NT, int nchunk, int SPC, int TPS, int parts,
                                                              const unsigned* __restrict__ amax, char* __restrict__ out) {
    cb_pack_entry(w, Cout, Cin, mode, nblk, NT, nchunk, SPC, TPS, parts, amax, out, (long)blockIdx.x * 256 + threadIdx.x);
}

// every layer of an encoder pass in one launch: blockIdx.y picks the layer
constexpr int CB_PACK_MAX = 16;
struct CbPackTable {
    const float* w[CB_PACK_MAX];
    const unsigned* amax[CB_PACK_MAX];
    char* out[CB_PACK_MAX];
    int Cout[CB_PACK_MAX], Cin[CB_PACK_MAX], mode[CB_PACK_MAX], nblk[CB_PACK_MAX], NT[CB_PACK_MAX], nchunk[CB_PACK_MAX],
        SPC[CB_PACK_MAX], TPS[CB_PACK_MAX];
    int parts;
};
__global__ __launch_bounds__(256) void cb_pack_weights_batch_kernel(CbPackTable t) {
    const int i = blockIdx.y;
    cb_pack_entry(t.w[i], t.Cout[i], t.Cin[i], t.mode[i], t.nblk[i], t.NT[i], t.nchunk[i], t.SPC[i], t.TPS[i], t.parts, t.amax[i],
                  t.out[i], (long)blockIdx.x * 256 + threadIdx.x);
}


// ---------------------------------------------------------------------------------------
// elementwise passes between the convolutions
// ---------------------------------------------------------------------------------------
constexpr float CB_IN_EPS = 1e-5f;       // nn.InstanceNorm2d default (function/VAE_CNN_NN.py:113)

// stats[plane] = {mean, 1/sqrt(biased var + eps)} of the H x W valid entries of a pitched plane; one wave per plane,
// two passes over registers (mean first, then squared deviations)
template <int EPT>
__global__ __launch_bounds__(256) void cb_plane_stats_kernel(const float* __restrict__ y, long nplanes, int H, int W,
                                                             int pitch, long plane_stride, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long plane = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= nplanes) return;
    const float* py = y + plane * plane_stride;
    const int Q = H * pitch;
    const unsigned mp = (unsigned)(0x100000000ull / (unsigned)pitch) + 1u;
    float v[EPT];
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = lane + 64 * e;
        const int r = (int)__umulhi((unsigned)i, mp), c = i - r * pitch;
        const bool ok = i < Q && c < W;
        v[e] = ok ? py[ok ? i : 0] : __builtin_nanf("");
        s += ok ? v[e] : 0.0f;
    }
    const float n = (float)(H * W);
    const float mean = wave_sum_all(s) / n;
    float q = 0.0f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const float d = v[e] - mean;
        q += (v[e] == v[e]) ? d * d : 0.0f;
    }
    const float var = wave_sum_all(q) / n;
    if (lane == 0) {
        stats[2 * plane] = mean;
        stats[2 * plane + 1] = 1.0f / sqrtf(var + CB_IN_EPS);
    }
}

// 8 values of one position -> 16 bytes of part 0 (bf16) or of parts 0 and 1 (fp16 parts of v * scale)
template <int PARTS>
__device__ __forceinline__ void cb_store8(const float (&v)[8], float scale, char* dst, long part_stride) {
    if constexpr (PARTS == 1) {
        cb_bf8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<cb_bf8*>(dst) = o;
    } else {
        cb_h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = v[e] * scale;
            hi[e] = (_Float16)t;
            lo[e] = (_Float16)(t - (float)hi[e]);
        }
        *reinterpret_cast<cb_h8*>(dst) = hi;
        *reinterpret_cast<cb_h8*>(dst + part_stride) = lo;
    }
}

struct CbNormFwd {
    const void* y;           // raw conv output [B][C][Qy], pitched rows (fp32, or fp16 for the kernels that stage it in LDS)
    const float* stats;      // [B * C][2]
    const float* skip;       // residual stream [B][C][H * W] or null
    float* stream_out;       // skip + norm(y), [B][C][H * W], or null
    char* act;               // channel-blocked output
    long part_stride;
    int B, C, H, W, pitch_y, Qy, relu;
    int mode;                // 0: reflection border 1, pitch W + 2;  1: four parity planes with a zero block border in front, pitch (W + 1) / 2 + 1
    int PP;                  // positions per (group-plane, sample)
    int opitch;              // pitch of the output planes
    float scale;
};

// a = [relu](norm(y)) (+ skip), written as the next convolution's operand.  One workgroup per (sample, 8 channels);
// a thread owns output positions (16 bytes each) and gathers their 8 source values, which consecutive lanes find at
// consecutive addresses of 8 planes.
template <int PARTS>
__global__ __launch_bounds__(256) void cb_norm_fwd_kernel(CbNormFwd p) {
    const int g8 = blockIdx.x, b = blockIdx.y;
    const int HW = p.H * p.W;
    float mean[8], rstd[8];
    const float* yp[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const long pl = (long)b * p.C + g8 * 8 + e;
        mean[e] = p.stats[2 * pl];
        rstd[e] = p.stats[2 * pl + 1];
        yp[e] = static_cast<const float*>(p.y) + pl * p.Qy;
    }
    const unsigned mop = (unsigned)(0x100000000ull / (unsigned)p.opitch) + 1u;
    if (p.mode == 1 && !p.skip && !p.stream_out) {
        // The four parity planes of a position take the four pixels of one 2 x 2 block: both pixels of a block row
        // come from one 8-byte load per channel, so every byte of y is requested once (two 4-byte loads at stride 8
        // relied on the first-level cache for the partner and fetched 1.4 times the planes).  A block that ends on
        // the last column loads one pixel to the left instead and takes the second half.
        for (int pos = threadIdx.x; pos < p.PP; pos += 256) {
            const int pr = (int)__umulhi((unsigned)pos, mop), pc = pos - pr * p.opitch;
            const int r0 = 2 * (pr - 1), c0 = 2 * (pc - 1);
            const bool inside = pr >= 1 && pc >= 1 && r0 < p.H && c0 < p.W;
            const bool pair = c0 + 1 < p.W;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int r = r0 + half;
                const bool ok = inside && r < p.H;
                const int src = ok ? r * p.pitch_y + (pair ? c0 : c0 - 1) : 0;
                float v[2][8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const cb_f2 t = *reinterpret_cast<const cb_f2u*>(yp[e] + src);
                    float t0 = ((pair ? t[0] : t[1]) - mean[e]) * rstd[e];
                    float t1 = (t[1] - mean[e]) * rstd[e];
                    if (p.relu) {
                        t0 = fmaxf(t0, 0.0f);
                        t1 = fmaxf(t1, 0.0f);
                    }
                    v[0][e] = ok ? t0 : 0.0f;
                    v[1][e] = ok && pair ? t1 : 0.0f;
                }
                if (p.act) {
#pragma unroll
                    for (int par = 0; par < 2; ++par) {
                        const long gp = (long)(g8 >> 1) * 8 + (half * 2 + par) * 2 + (g8 & 1);      // [chunk][plane][k half]
                        cb_store8<PARTS>(v[par], p.scale, p.act + ((gp * p.B + b) * p.PP + pos) * 16, p.part_stride);
                    }
                }
            }
        }
        return;
    }
    const int nplanes = p.mode == 1 ? 4 : 1;
    for (int pos = threadIdx.x; pos < p.PP; pos += 256) {
        const int pr = (int)__umulhi((unsigned)pos, mop), pc = pos - pr * p.opitch;
        for (int plane = 0; plane < nplanes; ++plane) {
            // group-plane index: [chunk][plane][k half]
            const long gp = p.mode == 1 ? (long)(g8 >> 1) * 8 + plane * 2 + (g8 & 1) : g8;
            char* dst = p.act + ((gp * p.B + b) * p.PP) * 16;
            int r, c;
            bool ok, interior = false;
            if (p.mode == 0) {
                r = pr - 1; c = pc - 1;
                ok = pr < p.H + 2 && pc < p.W + 2;
                interior = ok && r >= 0 && r < p.H && c >= 0 && c < p.W;
                r = r < 0 ? -r : (r >= p.H ? 2 * (p.H - 1) - r : r);
                c = c < 0 ? -c : (c >= p.W ? 2 * (p.W - 1) - c : c);
            } else {
                r = 2 * (pr - 1) + (plane >> 1); c = 2 * (pc - 1) + (plane & 1);
                ok = pr >= 1 && pc >= 1 && r < p.H && c < p.W;
            }
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.0f;
            if (ok) {
                const int src = r * p.pitch_y + c;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = (yp[e][src] - mean[e]) * rstd[e];
                    if (p.relu) t = fmaxf(t, 0.0f);
                    v[e] = t;
                }
                if (p.skip) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += p.skip[((long)b * p.C + g8 * 8 + e) * HW + r * p.W + c];
                }
                if (p.stream_out && (p.mode == 1 || interior)) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) p.stream_out[((long)b * p.C + g8 * 8 + e) * HW + r * p.W + c] = v[e];
                }
            }
            if (p.act) cb_store8<PARTS>(v, p.scale, dst + (long)pos * 16, p.part_stride);
        }
    }
}

struct CbNormBwd {
    const void* g;           // gradient w.r.t. this layer's activation, as the consumer's data gradient left it (fp32; bf16 for
                             // the kernels templated on its type)
    int gmode;               // 0: planes on this layer's grid, pitch_g;  1: reflection-padded grid (H+2) x (W+2) to fold, pitch_g;
                             // 2: four parity planes [B][4 C][Qg] of a stride-2 consumer, pitch_g;  3: none (base only)
    int pitch_g; long Qg;
    const float* base;       // [B][C][H * W] added to the gradient (the residual stream's gradient), or null
    float* sum_out;          // [B][C][H * W]: base + gradient, or null
    const void* y;           // raw conv output of this layer [B][C][Qy] (fp32, or fp16 for the kernels that stage it in LDS)
    const float* stats;
    int pitch_y; long Qy;
    int relu;
    int B, C, H, W;
    int omode;               // 0: zero border 2 (pitch W + 2, H + 4 rows);  1: zero border behind (pitch W + 1, H + 1 rows);  2: fp32 planes [B][C][H * W]
    char* dc;                // channel-blocked output (omode 0, 1) or float planes (omode 2)
    long part_stride;
    int PP, opitch;
    float scale;
    float* gsum;             // [B * C] plane sums of the result (conv bias gradient), or null
    const unsigned* amax_dc; // two fp16 parts: the record of the largest |dc| (a probe run of this pass); its scale replaces `scale`
    unsigned* amax_probe;    // probe run: nothing is written but the largest |dc| into this record
    int lds_gfloats;         // LDS-staged kernel: floats reserved for the gradient planes / the result's image
};

__device__ __forceinline__ float cb_fold_gather(const float* __restrict__ gp, int y, int x, int H, int W, int Wp) {
    const int r1 = (y == 1) ? 0 : ((y == H - 2) ? H + 1 : -1);
    const int r2 = (y == 1 && y == H - 2) ? H + 1 : -1;
    const int c1 = (x == 1) ? 0 : ((x == W - 2) ? W + 1 : -1);
    const int c2 = (x == 1 && x == W - 2) ? W + 1 : -1;
    const int rows[3] = {y + 1, r1, r2};
    const int cols[3] = {x + 1, c1, c2};
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb)
            if (rows[a] >= 0 && cols[bb] >= 0) s += gp[rows[a] * Wp + cols[bb]];
    return s;
}

// dc = rstd * (dn - mean(dn) - xh * mean(dn * xh)),  xh = (y - mean) * rstd,  dn = (base + gathered gradient) * [xh > 0 if relu]
// One workgroup per (sample, CH channels); a thread owns NPOS output positions, keeps dn and xh of their CH channels in
// registers between the plane reductions and the write.
template <int PARTS, int CH, int NPOS>
__global__ __launch_bounds__(256) void cb_norm_bwd_kernel(CbNormBwd p) {
    __shared__ float red[4][3 * CH];
    const int cg = blockIdx.x, b = blockIdx.y;
    const int HW = p.H * p.W;
    const float dc_scale = (PARTS == 2 && p.amax_dc) ? cb_pow2_scale(cb_slot_max(p.amax_dc)) : p.scale;
    unsigned smax = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ndom = p.omode == 2 ? HW : p.PP;
    const unsigned mop = (unsigned)(0x100000000ull / (unsigned)p.opitch) + 1u;
    float dn[NPOS][CH], xh[NPOS][CH];
    float s1[CH], s2[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) s1[e] = s2[e] = 0.0f;
    float mean[CH], rstd[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) {
        const long pl = (long)b * p.C + cg * CH + e;
        mean[e] = p.stats[2 * pl];
        rstd[e] = p.stats[2 * pl + 1];
    }
#pragma unroll
    for (int i = 0; i < NPOS; ++i) {
        const int pos = threadIdx.x + 256 * i;
        const int pr = (int)__umulhi((unsigned)pos, mop), pc = pos - pr * p.opitch;
        int r = pr, c = pc;
        if (p.omode == 0) { r = pr - 2; c = pc - 2; }
        const bool ok = pos < ndom && r >= 0 && r < p.H && c >= 0 && c < p.W;
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            dn[i][e] = 0.0f;
            xh[i][e] = 0.0f;
        }
        if (ok) {
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const long pl = (long)b * p.C + cg * CH + e;
                float gv = 0.0f;
                if (p.gmode == 0) gv = static_cast<const float*>(p.g)[pl * p.Qg + r * p.pitch_g + c];
                else if (p.gmode == 1) gv = cb_fold_gather(static_cast<const float*>(p.g) + pl * p.Qg, r, c, p.H, p.W, p.pitch_g);
                else if (p.gmode == 2)
                    gv = static_cast<const float*>(p.g)[((long)b * 4 * p.C + ((r & 1) * 2 + (c & 1)) * p.C + cg * CH + e) * p.Qg + (r >> 1) * p.pitch_g + (c >> 1)];
                if (p.base) gv += p.base[pl * HW + r * p.W + c];
                if (p.sum_out && !p.amax_probe) p.sum_out[pl * HW + r * p.W + c] = gv;
                const float x = (static_cast<const float*>(p.y)[pl * p.Qy + r * p.pitch_y + c] - mean[e]) * rstd[e];
                if (p.relu && !(x > 0.0f)) gv = 0.0f;
                dn[i][e] = gv;
                xh[i][e] = x;
                s1[e] += gv;
                s2[e] += gv * x;
            }
        }
    }
    // plane sums: waves in order, fixed tree inside a wave
#pragma unroll
    for (int e = 0; e < CH; ++e) {
        s1[e] = wave_sum_all(s1[e]);
        s2[e] = wave_sum_all(s2[e]);
    }
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            red[wave][e] = s1[e];
            red[wave][CH + e] = s2[e];
        }
    }
    __syncthreads();
    float m1[CH], m2[CH];
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int e = 0; e < CH; ++e) {
        m1[e] = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) * inv;
        m2[e] = ((red[0][CH + e] + red[1][CH + e]) + (red[2][CH + e] + red[3][CH + e])) * inv;
    }
    float gs[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) gs[e] = 0.0f;
    smax = 0;
#pragma unroll
    for (int i = 0; i < NPOS; ++i) {
        const int pos = threadIdx.x + 256 * i;
        if (pos >= ndom) continue;
        const int pr = (int)__umulhi((unsigned)pos, mop), pc = pos - pr * p.opitch;
        int r = pr, c = pc;
        if (p.omode == 0) { r = pr - 2; c = pc - 2; }
        const bool ok = r >= 0 && r < p.H && c >= 0 && c < p.W;
        float o[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            o[e] = ok ? rstd[e] * (dn[i][e] - m1[e] - xh[i][e] * m2[e]) : 0.0f;
            gs[e] += o[e];
            smax = max(smax, __float_as_uint(o[e]) & 0x7fffffffu);
        }
        if (p.amax_probe) continue;
        if (p.omode == 2) {
#pragma unroll
            for (int e = 0; e < CH; ++e)
                reinterpret_cast<float*>(p.dc)[((long)b * p.C + cg * CH + e) * HW + pos] = o[e];
        } else {
            if constexpr (CH == 8) {
                float o8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] = o[e];
                cb_store8<PARTS>(o8, dc_scale, p.dc + (((long)cg * p.B + b) * p.PP + pos) * 16, p.part_stride);
            }
        }
    }
    if (p.amax_probe) {
        cb_record_max(p.amax_probe, smax, (blockIdx.x + blockIdx.y * gridDim.x) * 4 + wave);
        return;
    }
    if (p.gsum) {
#pragma unroll
        for (int e = 0; e < CH; ++e) gs[e] = wave_sum_all(gs[e]);
        if (lane == 0) {
#pragma unroll
            for (int e = 0; e < CH; ++e) red[wave][2 * CH + e] = gs[e];
        }
        __syncthreads();
        if (threadIdx.x < CH) {
            const int e = threadIdx.x;
            p.gsum[(long)b * p.C + cg * CH + e] =
                (red[0][2 * CH + e] + red[1][2 * CH + e]) + (red[2][2 * CH + e] + red[3][2 * CH + e]);
        }
    }
}


// ---- the same two passes with their input planes staged in LDS -------------------------------
// The 8 planes a workgroup needs of a tensor [B][C][Q] are one contiguous run of 8 Q floats: they are copied into LDS
// by 16-byte LDS-DMA and every gather (reflection, parity planes, fold) becomes an LDS read.  The forward pass also
// takes the instance-norm statistics from the staged planes (mean first, then squared deviations), so the raw
// convolution output is read from memory once.

// nbytes (multiple of 16, 16-byte aligned source) -> LDS, all NT threads
template <int NT>
__device__ __forceinline__ void cb_dma_bytes(const char* __restrict__ src, char* dst, int nbytes) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int base = 0; base < nbytes; base += NT * 16) {
        const int e = base + wave * 1024 + lane * 16;
        if (e < nbytes)
            __builtin_amdgcn_global_load_lds(CB_GLOBAL_PTR(src + e), CB_LDS_PTR(dst + base + wave * 1024), 16, 0, 0);
    }
}
template <int NT>
__device__ __forceinline__ void cb_dma_floats(const float* __restrict__ src, float* dst, int n) {
    cb_dma_bytes<NT>(reinterpret_cast<const char*>(src), reinterpret_cast<char*>(dst), n * 4);
}

// ---- the LDS-staged passes, arithmetic by (channel, four positions) --------------------------
// The passes above spend their time on instructions, not on memory: a thread that owns a position reads eight
// 2- or 4-byte values from eight planes and carries eight-fold statistics.  Here thread t owns channel t & 7 and the
// position quads (t >> 3) + (NT / 8) k, k < KQ: one 8- or 16-byte LDS read brings four values of its plane, they stay
// in registers through the statistics (mean, then squared deviations: two plane reductions over the lanes of a channel)
// and the arithmetic, and leave as 16-bit values into an LDS image [position][8 channels] of the plane stack.  A
// second phase walks the output positions (reflection border, zero border or parity planes) and copies 16 bytes per
// position from that image.
template <int NT>
__device__ __forceinline__ float cb_chan_sum(float s, float* red) {       // sum over the threads of a channel (t & 7)
    s += __shfl_xor(s, 8);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane < 8) red[wave * 8 + lane] = s;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += red[w * 8 + (threadIdx.x & 7)];
    return t;
}

// four consecutive values of a plane in LDS (index a multiple of 4)
__device__ __forceinline__ void cb_load_quad(const float* pl, int i, float (&v)[4]) {
    const cb_f4 t = *reinterpret_cast<const cb_f4*>(pl + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = t[j];
}
typedef _Float16 cb_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void cb_load_quad(const _Float16* pl, int i, float (&v)[4]) {
    const cb_h4 t = *reinterpret_cast<const cb_h4*>(pl + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)t[j];
}

// one value into the image: bf16, or two fp16 parts of v * scale (second image `stride` bytes behind)
template <int PARTS>
__device__ __forceinline__ void cb_image_put(char* img, int stride, int pos, int e, float v, float scale) {
    if constexpr (PARTS == 1) {
        reinterpret_cast<__bf16*>(img)[pos * 8 + e] = (__bf16)v;
    } else {
        const float t = v * scale;
        const _Float16 hi = (_Float16)t;
        reinterpret_cast<_Float16*>(img)[pos * 8 + e] = hi;
        reinterpret_cast<_Float16*>(img + stride)[pos * 8 + e] = (_Float16)(t - (float)hi);
    }
}

// bit `i` of a mask as a predicate
__device__ __forceinline__ bool cb_bit(unsigned m, int i) { return (m >> i) & 1u; }

// STREAM (MODE 0): the residual stream is read (skip, may be null) and / or written (stream_out, may be null)
template <int PARTS, int MODE, bool STREAM, typename YT, int NT, int KQ>
__global__ __launch_bounds__(NT) void cb_norm_fwd_q_kernel(CbNormFwd p, float* __restrict__ stats_out) {
    static_assert(MODE == 0 || !STREAM, "the residual stream goes with the reflection-padded output");
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float* red = fsm;                                   // [64]
    YT* ybuf = reinterpret_cast<YT*>(fsm + 64);         // [8][Qy]
    const int Qv = p.H * p.pitch_y;                     // positions of the pitched plane that hold rows
    const int NQ = (Qv + 3) >> 2;
    char* img = reinterpret_cast<char*>(ybuf) + 8 * (size_t)p.Qy * sizeof(YT);
    const int img_stride = NQ * 64;
    const int g8 = blockIdx.x, b = blockIdx.y;
    const int HW = p.H * p.W;
    const long pl0 = (long)b * p.C + g8 * 8;
    const int e = threadIdx.x & 7, q0 = threadIdx.x >> 3;
    constexpr int QS = NT / 8;
    cb_dma_bytes<NT>(reinterpret_cast<const char*>(static_cast<const YT*>(p.y) + pl0 * p.Qy), reinterpret_cast<char*>(ybuf),
                     8 * p.Qy * (int)sizeof(YT));
    const unsigned mpy = (unsigned)(0x100000000ull / (unsigned)p.pitch_y) + 1u;
    const int gap = p.pitch_y - p.W;                    // pitched index - gap * row = index in the [H * W] planes
    // the residual stream's 8 planes [8][H * W] are one contiguous run in memory: staged in LDS like y, updated in
    // place by the threads that own the entries, and written back as one contiguous run
    float* sbuf = reinterpret_cast<float*>(img + (p.act ? PARTS * img_stride : 0));
    if (STREAM && p.skip) cb_dma_floats<NT>(p.skip + pl0 * HW, sbuf, 8 * HW);
    float* sbe = sbuf + e * HW;
    // which of this thread's entries are output pixels (okm), which quads are four pixels of one row (rowm)
    unsigned okm = 0, rowm = 0;
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int i0 = (q0 + QS * k) * 4;
        const int r0 = (int)__umulhi((unsigned)i0, mpy), c0 = i0 - r0 * p.pitch_y;
        rowm |= (i0 + 3 < Qv && c0 + 3 < p.W) ? 1u << k : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool wrap = c0 + j >= p.pitch_y;
            const bool ok = i0 + j < Qv && (wrap ? c0 + j - p.pitch_y : c0 + j) < p.W;
            okm |= ok ? 1u << (k * 4 + j) : 0u;
        }
    }
    cb_wait_vm<0>();
    __syncthreads();
    float v[KQ][4];
    float s = 0.0f;
    const YT* ype = ybuf + (size_t)e * p.Qy;
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int q = q0 + QS * k;
        cb_load_quad(ype, (q < NQ ? q : 0) * 4, v[k]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[k][j] = cb_bit(okm, k * 4 + j) ? v[k][j] : 0.0f;
            s += v[k][j];
        }
    }
    const float inv = 1.0f / (float)HW;
    const float mean = cb_chan_sum<NT>(s, red) * inv;
    s = 0.0f;
#pragma unroll
    for (int k = 0; k < KQ; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = cb_bit(okm, k * 4 + j) ? v[k][j] - mean : 0.0f;
            s = fmaf(d, d, s);
        }
    const float rstd = 1.0f / sqrtf(cb_chan_sum<NT>(s, red) * inv + CB_IN_EPS);
    if (threadIdx.x < 8) {
        stats_out[2 * (pl0 + e)] = mean;
        stats_out[2 * (pl0 + e) + 1] = rstd;
    }
    const float floor_ = p.relu ? 0.0f : -__builtin_huge_valf();
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int q = q0 + QS * k;
        const int i0 = q * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[k][j] = fmaxf((v[k][j] - mean) * rstd, floor_);
        if (STREAM) {
            const int r0 = (int)__umulhi((unsigned)i0, mpy), c0 = i0 - r0 * p.pitch_y;
            const int off0 = i0 - r0 * gap;
            if (cb_bit(rowm, k)) {
                if (p.skip) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[k][j] += sbe[off0 + j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) sbe[off0 + j] = v[k][j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cb_bit(okm, k * 4 + j)) {
                        const int off = off0 + j - (c0 + j >= p.pitch_y ? gap : 0);
                        if (p.skip) v[k][j] += sbe[off];
                        sbe[off] = v[k][j];
                    }
            }
        }
        if (p.act && q < NQ) {
#pragma unroll
            for (int j = 0; j < 4; ++j) cb_image_put<PARTS>(img, img_stride, i0 + j, e, v[k][j], p.scale);
        }
    }
    __syncthreads();
    if (STREAM && p.stream_out) {
        float* so = p.stream_out + pl0 * HW;
        for (int i = threadIdx.x * 4; i < 8 * HW; i += NT * 4) *reinterpret_cast<cb_f4*>(so + i) = *reinterpret_cast<const cb_f4*>(sbuf + i);
    }
    if (!p.act) return;
    const unsigned mop = (unsigned)(0x100000000ull / (unsigned)p.opitch) + 1u;
    constexpr int NPL = MODE == 1 ? 4 : 1;
#pragma unroll
    for (int plane = 0; plane < NPL; ++plane) {
        const long gp = MODE == 1 ? (long)(g8 >> 1) * 8 + plane * 2 + (g8 & 1) : g8;
        char* dst = p.act + ((gp * p.B + b) * p.PP) * 16;
        for (int pos = threadIdx.x; pos < p.PP; pos += NT) {
            const int pr = (int)__umulhi((unsigned)pos, mop), pc = pos - pr * p.opitch;
            int r, c;
            bool ok;
            if (MODE == 0) {
                r = pr - 1; c = pc - 1;
                ok = pr < p.H + 2 && pc < p.W + 2;
                r = r < 0 ? -r : (r >= p.H ? 2 * (p.H - 1) - r : r);
                c = c < 0 ? -c : (c >= p.W ? 2 * (p.W - 1) - c : c);
            } else {
                r = 2 * (pr - 1) + (plane >> 1); c = 2 * (pc - 1) + (plane & 1);
                ok = pr >= 1 && pc >= 1 && r < p.H && c < p.W;
            }
            const int src = ok ? r * p.pitch_y + c : 0;
            cb_u4 w0 = *reinterpret_cast<const cb_u4*>(img + src * 16);
            if (!ok) w0 = cb_u4{0u, 0u, 0u, 0u};
            *reinterpret_cast<cb_u4*>(dst + (long)pos * 16) = w0;
            if (PARTS == 2) {
                cb_u4 w1 = *reinterpret_cast<const cb_u4*>(img + img_stride + src * 16);
                if (!ok) w1 = cb_u4{0u, 0u, 0u, 0u};
                *reinterpret_cast<cb_u4*>(dst + p.part_stride + (long)pos * 16) = w1;
            }
        }
    }
}

template <int PARTS, int GMODE, int OMODE, bool BASE, typename YT, typename GT, int NT, int KQ>
__global__ __launch_bounds__(NT) void cb_norm_bwd_q_kernel(CbNormBwd p) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float* red = fsm;                                   // [64]
    YT* ybuf = reinterpret_cast<YT*>(fsm + 64);         // [8][Qy]
    const int Qy = (int)p.Qy, Qg = (int)p.Qg;
    float* gfl = fsm + 64 + 8 * Qy * (int)sizeof(YT) / 4;
    GT* gbuf = reinterpret_cast<GT*>(gfl);              // GMODE 0, 1: [8][Qg];  2: [4][8][Qg]
    char* img = reinterpret_cast<char*>(gbuf);          // the result's image takes the gradient's place once that is consumed
    const int Qv = p.H * p.pitch_y;
    const int NQ = (Qv + 3) >> 2;
    const int img_stride = NQ * 64;
    const int g8 = blockIdx.x, b = blockIdx.y;
    const int HW = p.H * p.W;
    const long pl0 = (long)b * p.C + g8 * 8;
    const int e = threadIdx.x & 7, q0 = threadIdx.x >> 3;
    constexpr int QS = NT / 8;
    const float dc_scale = (PARTS == 2 && p.amax_dc) ? cb_pow2_scale(cb_slot_max(p.amax_dc)) : p.scale;
    cb_dma_bytes<NT>(reinterpret_cast<const char*>(static_cast<const YT*>(p.y) + pl0 * Qy), reinterpret_cast<char*>(ybuf),
                     8 * Qy * (int)sizeof(YT));
    if (GMODE == 2) {
#pragma unroll
        for (int pl = 0; pl < 4; ++pl)
            cb_dma_bytes<NT>(reinterpret_cast<const char*>(static_cast<const GT*>(p.g) + (((long)b * 4 + pl) * p.C + g8 * 8) * Qg),
                             reinterpret_cast<char*>(gbuf + pl * 8 * Qg), 8 * Qg * (int)sizeof(GT));
    } else {
        cb_dma_bytes<NT>(reinterpret_cast<const char*>(static_cast<const GT*>(p.g) + pl0 * Qg), reinterpret_cast<char*>(gbuf),
                         8 * Qg * (int)sizeof(GT));
    }
    const unsigned mpy = (unsigned)(0x100000000ull / (unsigned)p.pitch_y) + 1u;
    const int gap = p.pitch_y - p.W;
    const float mean = p.stats[2 * (pl0 + e)], rstd = p.stats[2 * (pl0 + e) + 1];
    // the residual stream's gradient [8][H * W]: one contiguous run, staged in LDS, updated in place, written back whole
    float* bbuf = gfl + p.lds_gfloats;
    if (BASE) cb_dma_floats<NT>(p.base + pl0 * HW, bbuf, 8 * HW);
    float* bbe = bbuf + e * HW;
    const bool sums = BASE && p.sum_out && !p.amax_probe;
    unsigned okm = 0, rowm = 0;               // output pixels; quads that are four pixels of one row
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int i0 = (q0 + QS * k) * 4;
        const int r0 = (int)__umulhi((unsigned)i0, mpy), c0 = i0 - r0 * p.pitch_y;
        rowm |= (i0 + 3 < Qv && c0 + 3 < p.W) ? 1u << k : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool wrap = c0 + j >= p.pitch_y;
            const bool ok = i0 + j < Qv && (wrap ? c0 + j - p.pitch_y : c0 + j) < p.W;
            okm |= ok ? 1u << (k * 4 + j) : 0u;
        }
    }
    cb_wait_vm<0>();
    __syncthreads();
    float dn[KQ][4], xh[KQ][4];
    float s1 = 0.0f, s2 = 0.0f;
    const GT* ge = gbuf + e * Qg;
    const YT* ype = ybuf + (size_t)e * Qy;
    const float shift = -mean * rstd;
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int q = q0 + QS * k;
        const int i0 = q * 4;
        cb_load_quad(ype, (q < NQ ? q : 0) * 4, xh[k]);
        const int r0 = (int)__umulhi((unsigned)i0, mpy), c0 = i0 - r0 * p.pitch_y;
        int r = r0, c = c0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gv = 0.0f;
            if (cb_bit(okm, k * 4 + j)) {
                if (GMODE == 0) {
                    gv = (float)ge[r * p.pitch_g + c];
                } else if (GMODE == 1) {
                    // padded position (r + 1, c + 1) plus the mirrored border rows / columns that reflect onto (r, c)
                    const int Wp = p.pitch_g;
                    const int r1 = (r == 1) ? 0 : ((r == p.H - 2) ? p.H + 1 : -1);
                    const int c1 = (c == 1) ? 0 : ((c == p.W - 2) ? p.W + 1 : -1);
                    gv = (float)ge[(r + 1) * Wp + c + 1];
                    if (r1 >= 0) gv += (float)ge[r1 * Wp + c + 1];
                    if (c1 >= 0) gv += (float)ge[(r + 1) * Wp + c1];
                    if (r1 >= 0 && c1 >= 0) gv += (float)ge[r1 * Wp + c1];
                } else {
                    gv = (float)ge[((r & 1) * 2 + (c & 1)) * 8 * Qg + (r >> 1) * p.pitch_g + (c >> 1)];
                }
            }
            dn[k][j] = gv;
            if (++c == p.pitch_y) { c = 0; ++r; }
        }
        if (BASE) {                                      // base + gradient, before the mask
            const int off0 = i0 - r0 * gap;
            if (cb_bit(rowm, k)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dn[k][j] += bbe[off0 + j];
                if (sums) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bbe[off0 + j] = dn[k][j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cb_bit(okm, k * 4 + j)) {
                        const int off = off0 + j - (c0 + j >= p.pitch_y ? gap : 0);
                        dn[k][j] += bbe[off];
                        if (sums) bbe[off] = dn[k][j];
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = cb_bit(okm, k * 4 + j) ? fmaf(xh[k][j], rstd, shift) : 0.0f;
            float gv = dn[k][j];
            if (p.relu && !(x > 0.0f)) gv = 0.0f;
            dn[k][j] = gv;
            xh[k][j] = x;
            s1 += gv;
            s2 = fmaf(gv, x, s2);
        }
    }
    const float inv = 1.0f / (float)HW;
    const float m1 = cb_chan_sum<NT>(s1, red) * inv;      // (its barriers: every read of the gradient planes is done)
    const float m2 = cb_chan_sum<NT>(s2, red) * inv;
    if (sums) {
        float* so = p.sum_out + pl0 * HW;
        for (int i = threadIdx.x * 4; i < 8 * HW; i += NT * 4) *reinterpret_cast<cb_f4*>(so + i) = *reinterpret_cast<const cb_f4*>(bbuf + i);
    }
    float gs = 0.0f;
    unsigned omax = 0;
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int q = q0 + QS * k;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float o = cb_bit(okm, k * 4 + j) ? rstd * (dn[k][j] - m1 - xh[k][j] * m2) : 0.0f;
            gs += o;
            omax = max(omax, __float_as_uint(o) & 0x7fffffffu);
            dn[k][j] = o;
        }
        if (!p.amax_probe && q < NQ) {
#pragma unroll
            for (int j = 0; j < 4; ++j) cb_image_put<PARTS>(img, img_stride, q * 4 + j, e, dn[k][j], dc_scale);
        }
    }
    if (p.amax_probe) {
        cb_record_max(p.amax_probe, omax, (blockIdx.x + blockIdx.y * gridDim.x) * (NT / 64) + (threadIdx.x >> 6));
        return;
    }
    if (p.gsum) {
        gs = cb_chan_sum<NT>(gs, red);
        if (threadIdx.x < 8) p.gsum[pl0 + e] = gs;
    }
    __syncthreads();
    const unsigned mop = (unsigned)(0x100000000ull / (unsigned)p.opitch) + 1u;
    char* dst = p.dc + (((long)g8 * p.B + b) * p.PP) * 16;
    for (int pos = threadIdx.x; pos < p.PP; pos += NT) {
        const int pr = (int)__umulhi((unsigned)pos, mop), pc = pos - pr * p.opitch;
        const int r = OMODE == 0 ? pr - 2 : pr, c = OMODE == 0 ? pc - 2 : pc;
        const bool ok = r >= 0 && r < p.H && c >= 0 && c < p.W;
        const int src = ok ? r * p.pitch_y + c : 0;
        cb_u4 w0 = *reinterpret_cast<const cb_u4*>(img + src * 16);
        if (!ok) w0 = cb_u4{0u, 0u, 0u, 0u};
        *reinterpret_cast<cb_u4*>(dst + (long)pos * 16) = w0;
        if (PARTS == 2) {
            cb_u4 w1 = *reinterpret_cast<const cb_u4*>(img + img_stride + src * 16);
            if (!ok) w1 = cb_u4{0u, 0u, 0u, 0u};
            *reinterpret_cast<cb_u4*>(dst + p.part_stride + (long)pos * 16) = w1;
        }
    }
}

// The same backward pass for the layer whose planes are too large for LDS (the stem's: fp32 planes [H * W] in, fp32
// planes out, gradient gathered from the four parity planes of its stride-2 consumer).  One workgroup per plane; a
// thread owns KQ (row, four columns) items, loads them with 16- and 8-byte accesses that consecutive lanes make
// contiguous, keeps them in registers through the two plane sums, and stores 16 bytes per item.
// OT = __bf16: the result leaves rounded to bf16, planes PP entries apart (a multiple of 8, so that the stem's weight
// gradient -- which rounds this operand to bf16 in any case -- fetches 16 aligned bytes per 8 pixels).
template <int KQ, typename GT, typename OT>
__global__ __launch_bounds__(256) void cb_norm_bwd_plane_kernel(CbNormBwd p) {
    __shared__ float red[12];
    const int ch = blockIdx.x, b = blockIdx.y;
    const int HW = p.H * p.W;
    const long pl = (long)b * p.C + ch;
    const int wqw = p.W >> 2, tailn = p.W & 3;          // whole column quads per row, columns behind them
    const int nwhole = p.H * wqw, nitems = nwhole + (tailn ? p.H : 0);
    const unsigned mwq = (unsigned)(0x100000000ull / (unsigned)wqw) + 1u;
    const float* yp = static_cast<const float*>(p.y) + pl * p.Qy;
    const GT* gq = static_cast<const GT*>(p.g) + ((long)b * 4 * p.C + ch) * p.Qg;       // parity plane k: + k * C * Qg
    const int gps = p.C * (int)p.Qg;                    // (the host checks that four of these fit 31 bits)
    const float mean = p.stats[2 * pl], rstd = p.stats[2 * pl + 1];
    const float shift = -mean * rstd;
    // Items: first the whole quads (row, columns 4 q .. 4 q + 3), then one tail item per row for the W % 4 columns behind
    // them -- so that a wave holds items of one kind (but for one wave at the seam) and neither kind pays for the other's
    // branches.  Slots behind the last item hold zeros and add nothing to the sums.
    typedef GT g2u __attribute__((ext_vector_type(2), aligned(sizeof(GT))));
    float dn[KQ][4], xh[KQ][4];
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int it = threadIdx.x + 256 * k;
#pragma unroll
        for (int j = 0; j < 4; ++j) dn[k][j] = xh[k][j] = 0.0f;
        if (it < nwhole) {
            const int r = (int)__umulhi((unsigned)it, mwq), c = (it - r * wqw) * 4;
            const cb_f4 t = *reinterpret_cast<const cb_f4u*>(yp + (unsigned)(r * p.pitch_y + c));
            // columns c, c + 2 come from the even-column plane of this row's parity, c + 1, c + 3 from the odd one
            const GT* g0 = gq + (unsigned)(((r & 1) * 2) * gps + (r >> 1) * p.pitch_g + (c >> 1));
            const g2u ge = *reinterpret_cast<const g2u*>(g0), go = *reinterpret_cast<const g2u*>(g0 + gps);
#pragma unroll
            for (int j = 0; j < 4; ++j) xh[k][j] = fmaf(t[j], rstd, shift);
            dn[k][0] = (float)ge[0]; dn[k][1] = (float)go[0]; dn[k][2] = (float)ge[1]; dn[k][3] = (float)go[1];
        } else if (it < nitems) {
            const int r = it - nwhole, c = wqw * 4;
            const GT* g0 = gq + (unsigned)(((r & 1) * 2) * gps + (r >> 1) * p.pitch_g + (c >> 1));
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < tailn) {
                    xh[k][j] = fmaf(yp[(unsigned)(r * p.pitch_y + c + j)], rstd, shift);
                    dn[k][j] = (float)g0[(j & 1) * gps + (j >> 1)];
                }
        }
    }
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < KQ; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gv = dn[k][j];
            if (p.relu && !(xh[k][j] > 0.0f)) gv = 0.0f;
            dn[k][j] = gv;
            s1 += gv;
            s2 = fmaf(gv, xh[k][j], s2);
        }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    s1 = wave_sum_all(s1);
    s2 = wave_sum_all(s2);
    if (lane == 0) {
        red[wave] = s1;
        red[4 + wave] = s2;
    }
    __syncthreads();
    const float inv = 1.0f / (float)HW;
    const float m1 = ((red[0] + red[1]) + (red[2] + red[3])) * inv;
    const float m2 = ((red[4] + red[5]) + (red[6] + red[7])) * inv;
    float gs = 0.0f;
    constexpr bool O16 = !std::is_same<OT, float>::value;
    OT* op = reinterpret_cast<OT*>(p.dc) + pl * (O16 ? (long)p.PP : (long)HW);
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int it = threadIdx.x + 256 * k;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rstd * (dn[k][j] - m1 - xh[k][j] * m2);
        if (it < nwhole) {
            const int r = (int)__umulhi((unsigned)it, mwq), c = (it - r * wqw) * 4;
            gs += (o[0] + o[1]) + (o[2] + o[3]);
            if constexpr (O16) {
                *reinterpret_cast<cb_h4u*>(op + (unsigned)(r * p.W + c)) = cb_h4u{(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
            } else {
                *reinterpret_cast<cb_f4u*>(op + (unsigned)(r * p.W + c)) = cb_f4{o[0], o[1], o[2], o[3]};
            }
        } else if (it < nitems) {
            const int r = it - nwhole, c = wqw * 4;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < tailn) {
                    gs += o[j];
                    op[(unsigned)(r * p.W + c + j)] = (OT)o[j];
                }
        }
    }
    if (p.gsum) {
        gs = wave_sum_all(gs);
        if (lane == 0) red[8 + wave] = gs;
        __syncthreads();
        if (threadIdx.x == 0) p.gsum[pl] = (red[8] + red[9]) + (red[10] + red[11]);
    }
}

// ---------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------
// dW[tap][co][ci] = sum over samples and positions k of G[k][co] * act[plane(tap)][k + shift(tap)][ci]: the reduction
// runs over POSITIONS, which both channel-blocked operands have as their slow index, so both MFMA operands come out of
// LDS through the transposing read (ds_read_b64_tr_b16: a 16-lane group reads 4 positions x 16 channels and each lane
// receives 4 consecutive positions of its channel).  G is the output gradient in the zero-bordered layout the
// instance-norm backward pass writes (zeros wherever there is no output pixel), so border and scratch positions add
// nothing and a chunk of the reduction is simply KC consecutive positions; the activation slab of a chunk carries the
// span of the tap shifts as a halo.  v_mfma_f32_16x16x32 (bf16, or fp16 for two-part operands).  Two chunk buffers, the
// next chunk's LDS-DMA runs under the current chunk's MFMAs.  partial [Z][9][Cout][Cin], summed in fixed order by
// cb_wgrad_reduce_kernel.
typedef short cb_s4 __attribute__((ext_vector_type(4)));
typedef short cb_s8 __attribute__((ext_vector_type(8)));

struct CbWgrad {
    const char* g;            // output gradient, channel-blocked [Cout/8][B][PPg][8]
    const char* act;          // layer input, channel-blocked [group-plane][B][PPa][8]
    long g_part_stride, a_part_stride;
    float* partial;
    int B, Cout, Cin, PPg, PPa;
    int kbeg, kend;           // positions of G that take part (multiples of 32)
    int smin;                 // smallest tap shift (activation position = k + shift)
    int tap_off[9];           // shift(tap) - smin
    int tap_plane[9];
    int spg;                  // samples per z group
    float out_scale;
    const unsigned* amax_g;   // two fp16 parts: the record G was scaled by (undone in the result), or null
};

// A wave owns COT x 16 output channels x 16 input channels x 9 taps (the nine shifted activation fragments of a k step
// are each used for COT products).  Workgroup: 8 waves as WCO (co) x WCI (ci).  KC positions per chunk; LDS rows of a
// channel group are KC + 4 (G) and KC + 68 (activations: + up to 64 positions of tap span) positions long: = 4 mod 16, so
// the four groups a half-wave's transposed read touches fall on different banks.
template <int PARTS, int NPL, int COT, int WCO, int WCI, int KC>
__global__ __launch_bounds__(512) void cb_wgrad_kernel(CbWgrad p) {
    static_assert(WCO * WCI == 8 && KC % 64 == 0, "eight waves; chunks of whole 64-position blocks");
    constexpr int GSG = KC + 4, GSA = KC + 68;
    constexpr int GG = WCO * COT * 2;                 // channel groups of the G tile
    constexpr int GA = WCI * 2;                       // channel groups of the activation tile
    constexpr int GBUF = GG * GSG * 16;               // bytes: G tile of one chunk
    constexpr int ABUF = NPL * GA * GSA * 16;
    constexpr int PBUF = GBUF + ABUF;                 // one part of a chunk
    constexpr int BUF = PARTS * PBUF;
    constexpr int GBL = KC / 64, ABL = KC / 64 + 1;   // 64-position blocks per group
    constexpr int GP = GG * GBL, AP = NPL * GA * ABL; // 1 KiB pieces per chunk
    constexpr int NP = GP + AP;
    constexpr int PQ = NP / 8, PREM = NP % 8;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wa = wid / WCI, wb = wid - wa * WCI;    // co block (COT x 16) / ci tile (16)
    const int nct = (p.Cin + WCI * 16 - 1) / (WCI * 16);
    const int tco = blockIdx.x / nct, tci = blockIdx.x - tco * nct;
    const int co0 = tco * (WCO * COT * 16), ci0 = tci * (WCI * 16);
    const int b_begin = blockIdx.z * p.spg, b_end = min(p.B, b_begin + p.spg);
    const int ncs = (p.kend - p.kbeg + KC - 1) / KC;  // chunks per sample
    const int nchunks = (b_end - b_begin) * ncs;
    const int ngo = p.Cout >> 3, ngi = p.Cin >> 3;

    // staging plan: piece q = wid + 8 i; q < GP: G (group, block); else activation (plane, group, block)
    long src0[PQ + 1];
    int dst[PQ + 1], rel[PQ + 1], isg[PQ + 1];
#pragma unroll
    for (int i = 0; i < PQ + 1; ++i) {
        const int q = wid + 8 * i;
        src0[i] = 0; dst[i] = 0; rel[i] = 0; isg[i] = 1;
        if (q < GP) {
            const int gr = q / GBL, blk = q - gr * GBL;
            const int grp = min(co0 / 8 + gr, ngo - 1);
            src0[i] = (long)grp * p.B * p.PPg;
            dst[i] = (gr * GSG + blk * 64) * 16;
            rel[i] = blk * 64 + lane;
        } else if (q < NP) {
            const int a = q - GP;
            const int gg = a / ABL, blk = a - gg * ABL;
            const int pl = gg / GA, gr = gg - pl * GA;
            const int grp = min(ci0 / 8 + gr, ngi - 1);
            const long gpi = NPL == 4 ? (long)(grp >> 1) * 8 + pl * 2 + (grp & 1) : grp;
            src0[i] = gpi * p.B * p.PPa;
            dst[i] = GBUF + ((pl * GA + gr) * GSA + blk * 64) * 16;
            rel[i] = blk * 64 + lane + p.smin;
            isg[i] = 0;
        }
    }
    auto issue = [&](int ci, int buf) {
        const int b = b_begin + ci / ncs, kc = p.kbeg + (ci % ncs) * KC;
#pragma unroll
        for (int i = 0; i < PQ + 1; ++i) {
            if (i < PQ || wid < PREM) {
                const int PPx = isg[i] ? p.PPg : p.PPa;
                const int pos = min(max(kc + rel[i], 0), PPx - 1);
                const char* base = isg[i] ? p.g : p.act;
                const long pstride = isg[i] ? p.g_part_stride : p.a_part_stride;
#pragma unroll
                for (int pt = 0; pt < PARTS; ++pt)
                    __builtin_amdgcn_global_load_lds(CB_GLOBAL_PTR(base + pt * pstride + (src0[i] + (long)b * PPx + pos) * 16),
                                                     CB_LDS_PTR(smem + buf * BUF + pt * PBUF + dst[i]), 16, 0, 0);
            }
        }
    };

    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc[9][COT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < COT; ++c) acc[t][c] = f4{0.0f, 0.0f, 0.0f, 0.0f};

    // transposed-read addresses: lane = (k octet q4 = lane >> 4, li = lane & 15): li >> 2 = position within the block of
    // four, li & 3 = channel quad of the 16 channels
    const int q4 = lane >> 4, li = lane & 15;
    const int rq = li >> 2, cq = li & 3;
    int ga_addr[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c)
        ga_addr[c] = (((wa * COT + c) * 2 + (cq >> 1)) * GSG + 8 * q4 + rq) * 16 + 8 * (cq & 1);
    const int aa_addr = GBUF + ((wb * 2 + (cq >> 1)) * GSA + 8 * q4 + rq) * 16 + 8 * (cq & 1);
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = (p.tap_plane[t] * GA * GSA + p.tap_off[t]) * 16;

    auto tr = [&](const char* a) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cb_s4*)(a));
    };
    auto frag = [&](const char* a) {          // 8 consecutive positions of this lane's channel
        const cb_s4 lo = tr(a), hi = tr(a + 64);
        cb_s8 f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return __builtin_bit_cast(cb_bf8, f);
    };

    if (nchunks > 0) issue(0, 0);
    for (int ci = 0; ci < nchunks; ++ci) {
        cb_wait_vm<0>();
        __builtin_amdgcn_s_barrier();       // chunk ci has landed for everybody; everybody is done with the other buffer
        if (ci + 1 < nchunks) issue(ci + 1, (ci + 1) & 1);
        const char* buf = smem + (ci & 1) * BUF;
        const int kc = p.kbeg + (ci % ncs) * KC;
        const int nks = min(KC, p.kend - kc) >> 5;
        for (int ks = 0; ks < nks; ++ks) {
            cb_bf8 A[PARTS][COT];
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt)
#pragma unroll
                for (int c = 0; c < COT; ++c) A[pt][c] = frag(buf + pt * PBUF + ga_addr[c] + ks * 512);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                cb_bf8 Bv[PARTS];
#pragma unroll
                for (int pt = 0; pt < PARTS; ++pt) Bv[pt] = frag(buf + pt * PBUF + aa_addr + toff[t] + ks * 512);
#pragma unroll
                for (int c = 0; c < COT; ++c) {
                    if constexpr (PARTS == 2) {
                        // two fp16 parts per operand: high x low, low x high, high x high
                        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cb_h8, A[0][c]), __builtin_bit_cast(cb_h8, Bv[PARTS - 1]), acc[t][c], 0, 0, 0);
                        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cb_h8, A[PARTS - 1][c]), __builtin_bit_cast(cb_h8, Bv[0]), acc[t][c], 0, 0, 0);
                        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cb_h8, A[0][c]), __builtin_bit_cast(cb_h8, Bv[0]), acc[t][c], 0, 0, 0);
                    } else {
                        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0][c], Bv[0], acc[t][c], 0, 0, 0);
                    }
                }
            }
        }
    }
    // acc[t][c][reg]: row (co) = 4 (lane >> 4) + reg, column (ci) = lane & 15
    float* dstp = p.partial + (long)blockIdx.z * 9 * p.Cout * p.Cin;
    const float oscale = p.amax_g ? p.out_scale / cb_pow2_scale(cb_slot_max(p.amax_g)) : p.out_scale;
    const int cin = ci0 + wb * 16 + li;
    if (cin < p.Cin) {
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + (wa * COT + c) * 16 + 4 * q4 + r;
                    if (co < p.Cout) dstp[((long)t * p.Cout + co) * p.Cin + cin] = acc[t][c][r] * oscale;
                }
    }
}


// dW[n][c][tap] = sum_z partial[z][tap][n][c]   (fixed order)
__global__ __launch_bounds__(256) void cb_wgrad_reduce_kernel(const float* __restrict__ partial, int G, int Cout, int Cin,
                                                              float* __restrict__ dw) {
    const long total = 9L * Cout * Cin;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int z = 0;
    for (; z + 3 < G; z += 4) {
        s0 += partial[(long)z * total + idx];
        s1 += partial[(long)(z + 1) * total + idx];
        s2 += partial[(long)(z + 2) * total + idx];
        s3 += partial[(long)(z + 3) * total + idx];
    }
    for (; z < G; ++z) s0 += partial[(long)z * total + idx];
    const int c = (int)(idx % Cin);
    const long t = idx / Cin;
    const int n = (int)(t % Cout), tap = (int)(t / Cout);
    dw[((long)n * Cin + c) * 9 + tap] = (s0 + s1) + (s2 + s3);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static int cb_round_up(int v, int m) { return (v + m - 1) / m * m; }

// kind of product: 0 forward of a stride-1 reflection-padded layer, 1 forward of a stride-2 layer (four parity planes),
// 2 data gradient of a stride-1 layer, 3 data gradient of a stride-2 layer (2x2 taps, four parity planes out)
struct CbShape {
    int NT, TPS, SPC, NGP, NJ, pmode, nout, kch;
};
static bool cb_shape(int kind, int Cout, int Cin, int parts, CbShape& sh) {
    // two-part operands: one tap per stage (the stage ring holds both parts)
    const int tps9 = parts == 2 ? 1 : 3, spc9 = parts == 2 ? 9 : 3;
    const int tps4 = parts == 2 ? 1 : 2, spc4 = parts == 2 ? 4 : 2;
    if (kind == 0) { sh = {6, tps9, spc9, 2, 8, 0, Cout, Cin}; return true; }
    if (kind == 1 && Cout > 96) { sh = {6, tps9, spc9, 8, 5, 0, Cout, Cin}; return true; }
    if (kind == 1) { sh = {3, tps9, spc9, 8, 6, 0, Cout, Cin}; return true; }
    if (kind == 2) { sh = {6, tps9, spc9, 2, 8, 1, Cin, Cout}; return true; }
    if (kind == 3) { sh = {6, tps4, spc4, 2, 8, 2, 4 * Cin, Cout}; return true; }
    return false;
}

template <int PARTS, int NGP, int NJ, int WGM, int WGN, int WM, int WN, int TPS, int SPC, bool PERSIST, int NSW = 8, int NSB = 2, int OUT16 = 0>
static int cb_launch_conv(const CbConv& a, int nblk, hipStream_t st) {
    constexpr int NT = WGN * WN;
    constexpr size_t lds_main = (size_t)CB_NBUF * TPS * NT * PARTS * 1024 + (size_t)NSB * PARTS * NGP * NJ * 64 * 16;
    constexpr size_t lds_epi = (size_t)WGN * 32 * (128 + 4) * 4;                // the epilogue's transpose buffer
    constexpr size_t lds = PERSIST ? lds_main + lds_epi : (lds_main > lds_epi ? lds_main : lds_epi);
    static_assert(lds <= 160 * 1024, "stage ring + two slabs (+ the epilogue buffer) must fit the LDS");
    auto kern = cb_conv_kernel<PARTS, NGP, NJ, WGM, WGN, WM, WN, TPS, SPC, PERSIST, NSW, NSB, OUT16>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long rows = (long)a.B * a.Qr;
    const int ntiles = (int)((rows + CB_ROWS - 1) / CB_ROWS) * nblk;
    int grid = ntiles;
    if (PERSIST) {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (grid > cus) grid = cus;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a, nblk, ntiles);
    MV_LAUNCH_CHECK("cb_conv_kernel");
    return 0;
}

template <int PARTS, int NPL, int COT, int WCO, int WCI, int KC>
static int cb_launch_wgrad(const CbWgrad& a, int ntiles, int Gz, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * PARTS * ((size_t)WCO * COT * 2 * (KC + 4) * 16 + (size_t)NPL * WCI * 2 * (KC + 68) * 16);
    static_assert(lds <= 160 * 1024, "two chunk buffers must fit the LDS");
    auto kern = cb_wgrad_kernel<PARTS, NPL, COT, WCO, WCI, KC>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(ntiles, 1, Gz), dim3(512), lds, st, a);
    MV_LAUNCH_CHECK("cb_wgrad_kernel");
    return 0;
}

// tile shapes of the weight gradient: one plane: 96 co x 64 ci, chunks of 128 positions; four parity planes: 192 co x
// 32 ci, chunks of 64 (sixteen group-planes of activations per chunk)
static void cb_wgrad_tiles(int planes, int parts, int& tco, int& tci) {
    tco = planes == 4 ? (parts == 2 ? 128 : 192) : 96;
    tci = planes == 4 ? (parts == 2 ? 16 : 32) : 64;      // two parts of sixteen group-planes: one ci tile per workgroup fits
}
static int cb_wgrad_groups(int B, int ntiles) {
    int G = 256 / ntiles;
    if (G < 1) G = 1;
    if (G > B) G = B;
    return G;
}

}  // namespace mdnvae
using namespace mdnvae;

extern "C" {

size_t mdnvae_cb_packed_bytes(int Cout, int Cin, int kind, int parts) {
    CbShape sh;
    if (parts < 1 || parts > 2 || !cb_shape(kind, Cout, Cin, parts, sh) || sh.kch % 16 || Cout % 8 || Cin % 8) return 0;
    const int nblk = (sh.nout + sh.NT * 32 - 1) / (sh.NT * 32);
    return (size_t)nblk * (sh.kch / 16) * sh.SPC * sh.TPS * sh.NT * parts * 1024;
}

int mdnvae_cb_pack_weights(const float* w, int Cout, int Cin, int kind, int parts, const unsigned* amax, void* packed,
                           void* stream) {
    MV_REQUIRE(w && packed, E_ARG, "mdnvae_cb_pack_weights: null pointer");
    const size_t bytes = mdnvae_cb_packed_bytes(Cout, Cin, kind, parts);
    MV_REQUIRE(bytes > 0, E_SHAPE, "mdnvae_cb_pack_weights: unsupported shape (%d -> %d, kind %d, parts %d)", Cin, Cout, kind, parts);
    MV_REQUIRE(parts == 1 || amax, E_ARG, "mdnvae_cb_pack_weights: two fp16 parts need the tensor's largest magnitude");
    CbShape sh;
    cb_shape(kind, Cout, Cin, parts, sh);
    const int nblk = (sh.nout + sh.NT * 32 - 1) / (sh.NT * 32);
    const long total = (long)(bytes / 2);
    hipLaunchKernelGGL(cb_pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       Cout, Cin, sh.pmode, nblk, sh.NT, sh.kch / 16, sh.SPC, sh.TPS, parts, amax, static_cast<char*>(packed));
    MV_LAUNCH_CHECK("cb_pack_weights_kernel");
    return 0;
}

int mdnvae_cb_pack_weights_batch(int n, const float* const* w, const int* Cout, const int* Cin, const int* kind, int parts,
                                 const unsigned* const* amax, void* const* packed, void* stream) {
    MV_REQUIRE(n >= 0 && n <= CB_PACK_MAX, E_ARG, "mdnvae_cb_pack_weights_batch: %d layers (at most %d)", n, CB_PACK_MAX);
    if (n == 0) return 0;
    MV_REQUIRE(w && Cout && Cin && kind && packed, E_ARG, "mdnvae_cb_pack_weights_batch: null table");
    MV_REQUIRE(parts == 1 || amax, E_ARG, "mdnvae_cb_pack_weights_batch: two fp16 parts need the tensors' largest magnitudes");
    CbPackTable t{};
    t.parts = parts;
    long most = 0;
    for (int i = 0; i < n; ++i) {
        MV_REQUIRE(w[i] && packed[i] && (parts == 1 || amax[i]), E_ARG, "mdnvae_cb_pack_weights_batch: null pointer (layer %d)", i);
        const size_t bytes = mdnvae_cb_packed_bytes(Cout[i], Cin[i], kind[i], parts);
        MV_REQUIRE(bytes > 0, E_SHAPE, "mdnvae_cb_pack_weights_batch: unsupported shape (%d -> %d, kind %d, parts %d)", Cin[i], Cout[i],
                   kind[i], parts);
        CbShape sh;
        cb_shape(kind[i], Cout[i], Cin[i], parts, sh);
        t.w[i] = w[i]; t.amax[i] = amax ? amax[i] : nullptr; t.out[i] = static_cast<char*>(packed[i]);
        t.Cout[i] = Cout[i]; t.Cin[i] = Cin[i]; t.mode[i] = sh.pmode;
        t.nblk[i] = (sh.nout + sh.NT * 32 - 1) / (sh.NT * 32);
        t.NT[i] = sh.NT; t.nchunk[i] = sh.kch / 16; t.SPC[i] = sh.SPC; t.TPS[i] = sh.TPS;
        if ((long)(bytes / 2) > most) most = (long)(bytes / 2);
    }
    hipLaunchKernelGGL(cb_pack_weights_batch_kernel, dim3((unsigned)((most + 255) / 256), n), dim3(256), 0, (hipStream_t)stream, t);
    MV_LAUNCH_CHECK("cb_pack_weights_batch_kernel");
    return 0;
}

int mdnvae_cb_conv(const void* act, long part_stride, const void* packed_w, const float* bias, void* out, int out16, int B,
                   int Cout, int Cin, int kind, int rows, int pitch, int PP, int parts, float out_scale,
                   const unsigned* absmax_act, const unsigned* absmax_w, void* stream) {
    MV_REQUIRE(B >= 0, E_ARG, "mdnvae_cb_conv: negative batch");
    if (B == 0) return 0;
    MV_REQUIRE(act && packed_w && out, E_ARG, "mdnvae_cb_conv: null pointer");
    MV_REQUIRE(parts == 1 || parts == 2, E_ARG, "mdnvae_cb_conv: parts = %d (1 or 2)", parts);
    CbShape sh;
    MV_REQUIRE(cb_shape(kind, Cout, Cin, parts, sh) && sh.kch % 16 == 0, E_SHAPE, "mdnvae_cb_conv: unsupported product (%d -> %d, kind %d)",
               Cin, Cout, kind);
    MV_REQUIRE(parts == 1 || (absmax_w && !out16), E_ARG, "mdnvae_cb_conv: two-part operands need the weights' record and fp32 output");
    MV_REQUIRE(rows >= 1 && pitch >= 2, E_SHAPE, "mdnvae_cb_conv: bad grid %d x %d", rows, pitch);
    CbConv a{};
    a.act = static_cast<const char*>(act);
    a.part_stride = part_stride;
    a.w = static_cast<const char*>(packed_w);
    a.bias = bias;
    a.out = static_cast<float*>(out);
    a.B = B; a.PP = PP;
    a.nchunk = sh.kch / 16;
    a.Qr = cb_round_up(rows * pitch, 8);
    a.ntot = sh.nout;
    MV_REQUIRE(out16 >= 0 && out16 <= 2 && (out16 != 1 || kind <= 1) && (out16 != 2 || kind >= 2), E_ARG,
               "mdnvae_cb_conv: out16 = %d (1: fp16, the forward products; 2: bf16, the data gradients)", out16);
    a.out_scale = out_scale;
    a.amax_a = absmax_act; a.amax_w = absmax_w;
    const int slabpos = sh.NJ * 64;
    int ntaps = 0;
    if (kind == 0 || kind == 2) {
        for (int t = 0; t < 9; ++t) a.tap_lds[ntaps++] = ((t / 3) * pitch + (t % 3)) * 16;
        a.halo = 2 * pitch + 2;
    } else if (kind == 1) {
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t % 3;
            const int plane = (ky == 1 ? 0 : 1) * 2 + (kx == 1 ? 0 : 1);
            a.tap_lds[ntaps++] = (plane * 2 * slabpos + (ky > 0) * pitch + (kx > 0)) * 16;
        }
        a.halo = pitch + 1;
    } else {
        for (int t = 0; t < 4; ++t) a.tap_lds[ntaps++] = ((t >> 1) * pitch + (t & 1)) * 16;
        a.halo = pitch + 1;
    }
    MV_REQUIRE(a.Qr >= CB_ROWS, E_SHAPE, "mdnvae_cb_conv: %d rows per sample are fewer than the %d-row tile", a.Qr, CB_ROWS);
    MV_REQUIRE(CB_ROWS + 2 * a.halo <= slabpos, E_SHAPE, "mdnvae_cb_conv: pitch %d needs a slab of %d positions (%d fit)", pitch,
               CB_ROWS + 2 * a.halo, slabpos);
    MV_REQUIRE(PP >= a.Qr + a.halo, E_SHAPE, "mdnvae_cb_conv: input planes of %d positions are shorter than %d", PP, a.Qr + a.halo);
    const int nblk = (sh.nout + sh.NT * 32 - 1) / (sh.NT * 32);
    hipStream_t st = (hipStream_t)stream;
    if (parts == 2) {
        if (kind == 0 || kind == 2) return cb_launch_conv<2, 2, 8, 4, 2, 2, 3, 1, 9, true, 8, 2, 0>(a, nblk, st);
        if (kind == 3) return cb_launch_conv<2, 2, 8, 4, 2, 2, 3, 1, 4, true, 8, 2, 0>(a, nblk, st);
        if (sh.NT == 6) return cb_launch_conv<2, 8, 5, 4, 2, 2, 3, 1, 9, false, 8, 1, 0>(a, nblk, st);
        return cb_launch_conv<2, 8, 6, 8, 1, 1, 3, 1, 9, false, 8, 1, 0>(a, nblk, st);
    }
    if (kind == 0 && out16) return cb_launch_conv<1, 2, 8, 4, 2, 2, 3, 3, 3, true, 8, 2, 1>(a, nblk, st);
    if (kind == 2 && out16) return cb_launch_conv<1, 2, 8, 4, 2, 2, 3, 3, 3, true, 8, 2, 2>(a, nblk, st);
    if (kind == 0 || kind == 2) return cb_launch_conv<1, 2, 8, 4, 2, 2, 3, 3, 3, true, 8, 2, 0>(a, nblk, st);
    if (kind == 3 && out16) return cb_launch_conv<1, 2, 8, 4, 2, 2, 3, 2, 2, true, 8, 2, 2>(a, nblk, st);
    if (kind == 3) return cb_launch_conv<1, 2, 8, 4, 2, 2, 3, 2, 2, true, 8, 2, 0>(a, nblk, st);
    if (sh.NT == 6) {
        if (out16) return cb_launch_conv<1, 8, 5, 4, 2, 2, 3, 3, 3, false, 8, 2, 1>(a, nblk, st);
        return cb_launch_conv<1, 8, 5, 4, 2, 2, 3, 3, 3, false, 8, 2, 0>(a, nblk, st);
    }
    if (out16) return cb_launch_conv<1, 8, 6, 8, 1, 1, 3, 3, 3, false, 8, 2, 1>(a, nblk, st);
    return cb_launch_conv<1, 8, 6, 8, 1, 1, 3, 3, 3, false, 8, 2, 0>(a, nblk, st);
}

int mdnvae_cb_plane_stats(const float* y, long nplanes, int H, int W, int pitch, long plane_stride, float* stats,
                          void* stream) {
    MV_REQUIRE(nplanes >= 0 && H >= 1 && W >= 1 && pitch >= W, E_ARG, "mdnvae_cb_plane_stats: bad size");
    if (nplanes == 0) return 0;
    MV_REQUIRE(y && stats, E_ARG, "mdnvae_cb_plane_stats: null pointer");
    const int Q = H * pitch;
    MV_REQUIRE(Q <= 2048, E_SHAPE, "mdnvae_cb_plane_stats: plane of %d positions exceeds 2048", Q);
    dim3 grid((unsigned)((nplanes + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (Q <= 640) hipLaunchKernelGGL(cb_plane_stats_kernel<10>, grid, dim3(256), 0, st, y, nplanes, H, W, pitch, plane_stride, stats);
    else hipLaunchKernelGGL(cb_plane_stats_kernel<32>, grid, dim3(256), 0, st, y, nplanes, H, W, pitch, plane_stride, stats);
    MV_LAUNCH_CHECK("cb_plane_stats_kernel");
    return 0;
}

int mdnvae_cb_norm_fwd(const void* y, int y16, float* stats, int compute_stats, const float* skip, float* stream_out,
                       void* act, long part_stride, int B, int C, int H, int W, int pitch_y, long Qy, int relu, int mode,
                       int PP, int parts, float scale, void* stream) {
    MV_REQUIRE(B >= 0 && C >= 8 && C % 8 == 0 && H >= 2 && W >= 2, E_SHAPE, "mdnvae_cb_norm_fwd: bad shape");
    if (B == 0) return 0;
    MV_REQUIRE(y && stats && (act || stream_out), E_ARG, "mdnvae_cb_norm_fwd: null pointer");
    MV_REQUIRE(mode == 0 || mode == 1, E_ARG, "mdnvae_cb_norm_fwd: mode %d", mode);
    MV_REQUIRE(parts == 1 || parts == 2, E_ARG, "mdnvae_cb_norm_fwd: parts = %d", parts);
    MV_REQUIRE(B <= 65535, E_SHAPE, "mdnvae_cb_norm_fwd: batch %d exceeds the grid limit", B);
    MV_REQUIRE(mode == 0 || (!skip && !stream_out), E_ARG, "mdnvae_cb_norm_fwd: skip / stream_out go with mode 0");
    MV_REQUIRE(!y16 || compute_stats, E_ARG, "mdnvae_cb_norm_fwd: fp16 planes go with compute_stats = 1");
    CbNormFwd a{};
    a.y = y; a.stats = stats; a.skip = skip; a.stream_out = stream_out;
    a.act = static_cast<char*>(act);
    a.part_stride = part_stride;
    a.B = B; a.C = C; a.H = H; a.W = W; a.pitch_y = pitch_y; a.Qy = (int)Qy; a.relu = relu; a.mode = mode; a.PP = PP;
    a.opitch = mode == 0 ? W + 2 : (W + 1) / 2 + 1;
    const int orows = mode == 0 ? H + 2 : (H + 1) / 2 + 1;
    MV_REQUIRE(!act || PP >= orows * a.opitch, E_SHAPE, "mdnvae_cb_norm_fwd: %d positions per plane are fewer than %d x %d", PP, orows, a.opitch);
    a.scale = scale;
    dim3 grid(C / 8, B);
    hipStream_t st = (hipStream_t)stream;
    if (compute_stats) {
        const int NQ = (H * pitch_y + 3) / 4;
        const bool streams = skip || stream_out;
        const size_t lds = (size_t)64 * 4 + 8 * (size_t)Qy * (y16 ? 2 : 4) + (act ? (size_t)parts * NQ * 64 : 0) +
                           (streams ? (size_t)32 * H * W : 0);
        MV_REQUIRE(Qy % (y16 ? 8 : 4) == 0 && Qy >= (long)H * pitch_y && pitch_y >= 4 && lds <= 160 * 1024 && NQ <= 512, E_SHAPE,
                   "mdnvae_cb_norm_fwd: statistics in the pass need planes of a multiple of %d positions that fit the LDS (got %ld)",
                   y16 ? 8 : 4, Qy);
#define MV_FWD(P_, M_, S_, T_, NT_, KQ_)                                                                                      \
    do {                                                                                                                      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cb_norm_fwd_q_kernel<P_, M_, S_, T_, NT_, KQ_>),              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                    \
        hipLaunchKernelGGL((cb_norm_fwd_q_kernel<P_, M_, S_, T_, NT_, KQ_>), grid, dim3(NT_), lds, st, a, stats);             \
    } while (0)
#define MV_FWD_S(P_, M_, S_, T_)                          \
    do {                                                  \
        if (NQ <= 160) MV_FWD(P_, M_, S_, T_, 256, 5);    \
        else MV_FWD(P_, M_, S_, T_, 512, 8);              \
    } while (0)
#define MV_FWD_T(P_, M_, S_)                       \
    do {                                           \
        if (y16) MV_FWD_S(P_, M_, S_, _Float16);   \
        else MV_FWD_S(P_, M_, S_, float);          \
    } while (0)
        if (parts == 1 && mode == 0 && streams) MV_FWD_T(1, 0, true);
        else if (parts == 1 && mode == 0) MV_FWD_T(1, 0, false);
        else if (parts == 1) MV_FWD_T(1, 1, false);
        else if (mode == 0 && streams) MV_FWD_T(2, 0, true);
        else if (mode == 0) MV_FWD_T(2, 0, false);
        else MV_FWD_T(2, 1, false);
#undef MV_FWD_T
#undef MV_FWD_S
#undef MV_FWD
        MV_LAUNCH_CHECK("cb_norm_fwd_q_kernel");
        return 0;
    }
    if (parts == 1) hipLaunchKernelGGL(cb_norm_fwd_kernel<1>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(cb_norm_fwd_kernel<2>, grid, dim3(256), 0, st, a);
    MV_LAUNCH_CHECK("cb_norm_fwd_kernel");
    return 0;
}

int mdnvae_cb_norm_bwd(const void* g, int gmode, int pitch_g, long Qg, const float* base, float* sum_out, const void* y,
                       int y16, const float* stats, int pitch_y, long Qy, int relu, int B, int C, int H, int W, int omode,
                       void* dc, long part_stride, int PP, int parts, float scale, float* gsum, const unsigned* absmax_dc,
                       unsigned* absmax_probe, int g16, void* stream) {
    MV_REQUIRE(B >= 0 && C >= 1 && H >= 3 && W >= 3, E_SHAPE, "mdnvae_cb_norm_bwd: bad shape");
    if (B == 0) return 0;
    MV_REQUIRE(y && stats && (dc || absmax_probe) && (g || gmode == 3), E_ARG, "mdnvae_cb_norm_bwd: null pointer");
    MV_REQUIRE(gmode >= 0 && gmode <= 3 && omode >= 0 && omode <= 3, E_ARG, "mdnvae_cb_norm_bwd: gmode %d / omode %d", gmode, omode);
    MV_REQUIRE(gmode != 3 || base, E_ARG, "mdnvae_cb_norm_bwd: no gradient source");
    MV_REQUIRE(parts == 1 || parts == 2, E_ARG, "mdnvae_cb_norm_bwd: parts = %d", parts);
    MV_REQUIRE(B <= 65535, E_SHAPE, "mdnvae_cb_norm_bwd: batch %d exceeds the grid limit", B);
    MV_REQUIRE(g16 == 0 || (g16 == 2 && parts == 1), E_ARG, "mdnvae_cb_norm_bwd: g16 = %d (0: fp32; 2: bf16, with bf16 operands out)", g16);
    CbNormBwd a{};
    a.g = g; a.gmode = gmode; a.pitch_g = pitch_g; a.Qg = Qg; a.base = base; a.sum_out = sum_out;
    a.y = y; a.stats = stats; a.pitch_y = pitch_y; a.Qy = Qy; a.relu = relu;
    a.B = B; a.C = C; a.H = H; a.W = W; a.omode = omode;
    a.dc = static_cast<char*>(dc);
    a.part_stride = part_stride; a.PP = PP; a.scale = scale; a.gsum = gsum;
    a.amax_dc = absmax_dc; a.amax_probe = absmax_probe;
    MV_REQUIRE(parts == 1 || omode >= 2 || absmax_dc || absmax_probe, E_ARG,
               "mdnvae_cb_norm_bwd: two fp16 parts need the record of the result's largest magnitude (a probe run)");
    a.opitch = omode == 0 ? W + 2 : (omode == 1 ? W + 1 : W);
    hipStream_t st = (hipStream_t)stream;
    if (omode >= 2) {
        MV_REQUIRE(!y16, E_ARG, "mdnvae_cb_norm_bwd: plain planes out take fp32 planes in");
        MV_REQUIRE(omode == 2 || (g16 == 2 && PP >= H * W && PP % 8 == 0), E_ARG,
                   "mdnvae_cb_norm_bwd: bf16 planes out go with bf16 gradient planes in and a plane pitch (PP = %d) that is a "
                   "multiple of 8 and holds the %d x %d plane", PP, H, W);
        MV_REQUIRE(H * W <= 32 * 256, E_SHAPE, "mdnvae_cb_norm_bwd: plane of %d entries exceeds %d", H * W, 32 * 256);
        const int items = H * ((W + 3) / 4);
        if (gmode == 2 && !base && !sum_out && items <= 8 * 256 && W >= 4 && 4L * C * Qg < (1L << 31) && pitch_g >= (W + 1) / 2 + 1 && Qg >= (long)((H + 1) / 2) * pitch_g) {
            if (omode == 3) {
                if (items <= 5 * 256) hipLaunchKernelGGL((cb_norm_bwd_plane_kernel<5, __bf16, __bf16>), dim3(C, B), dim3(256), 0, st, a);
                else hipLaunchKernelGGL((cb_norm_bwd_plane_kernel<8, __bf16, __bf16>), dim3(C, B), dim3(256), 0, st, a);
            } else if (g16) {
                if (items <= 5 * 256) hipLaunchKernelGGL((cb_norm_bwd_plane_kernel<5, __bf16, float>), dim3(C, B), dim3(256), 0, st, a);
                else hipLaunchKernelGGL((cb_norm_bwd_plane_kernel<8, __bf16, float>), dim3(C, B), dim3(256), 0, st, a);
            } else {
                if (items <= 5 * 256) hipLaunchKernelGGL((cb_norm_bwd_plane_kernel<5, float, float>), dim3(C, B), dim3(256), 0, st, a);
                else hipLaunchKernelGGL((cb_norm_bwd_plane_kernel<8, float, float>), dim3(C, B), dim3(256), 0, st, a);
            }
            MV_LAUNCH_CHECK("cb_norm_bwd_plane_kernel");
            return 0;
        }
        MV_REQUIRE(!g16, E_SHAPE, "mdnvae_cb_norm_bwd: bf16 gradient planes need the plane kernel's shapes");
        hipLaunchKernelGGL((cb_norm_bwd_kernel<1, 1, 32>), dim3(C, B), dim3(256), 0, st, a);
        MV_LAUNCH_CHECK("cb_norm_bwd_kernel");
        return 0;
    }
    MV_REQUIRE(C % 8 == 0, E_SHAPE, "mdnvae_cb_norm_bwd: %d channels are not a multiple of 8", C);
    const int orows = omode == 0 ? H + 4 : H + 1;
    MV_REQUIRE(PP >= orows * a.opitch, E_SHAPE, "mdnvae_cb_norm_bwd: %d positions per plane are fewer than %d x %d", PP, orows, a.opitch);
    MV_REQUIRE(PP <= 9 * 256, E_SHAPE, "mdnvae_cb_norm_bwd: %d positions per plane exceed %d", PP, 9 * 256);
    dim3 grid(C / 8, B);
    // planes staged in LDS when they fit and their runs of 8 planes are 16-byte aligned
    const int NQ = (H * pitch_y + 3) / 4;
    const size_t gfl = gmode == 3 ? 0 : (gmode == 2 ? (size_t)32 * Qg : (size_t)8 * Qg);
    const size_t gbytes = (gfl * (g16 ? 2 : 4) + 15) / 16 * 16;
    const size_t img = (size_t)parts * NQ * 64;
    a.lds_gfloats = (int)((gbytes > img ? gbytes : img) / 4);
    const size_t lds = (size_t)64 * 4 + 8 * (size_t)Qy * (y16 ? 2 : 4) + (size_t)a.lds_gfloats * 4 + (base ? (size_t)32 * H * W : 0);
    const bool aligned = Qy % (y16 ? 8 : 4) == 0 && Qy >= (long)H * pitch_y && pitch_y >= 4;      // (runs of 8 planes of g: 32 Qg bytes)
    const bool covered = gmode == 0 ? Qg >= (long)H * pitch_g : (gmode == 1 ? Qg >= (long)(H + 2) * pitch_g && pitch_g >= W + 2
                                                                            : Qg >= (long)((H + 1) / 2) * pitch_g);
    if (aligned && covered && lds <= 160 * 1024 && NQ <= 512 && H >= 4 && W >= 4 && (!sum_out || base) && gmode != 3 &&
        !(parts == 2 && y16) && (!g16 || y16)) {
#define MV_BWD(P_, G_, O_, B_, T_, GT_, NT_, KQ_)                                                                             \
    do {                                                                                                                      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cb_norm_bwd_q_kernel<P_, G_, O_, B_, T_, GT_, NT_, KQ_>),     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                    \
        hipLaunchKernelGGL((cb_norm_bwd_q_kernel<P_, G_, O_, B_, T_, GT_, NT_, KQ_>), grid, dim3(NT_), lds, st, a);           \
    } while (0)
#define MV_BWD_T(G_, O_, B_, NT_, KQ_)                                         \
    do {                                                                       \
        if (parts == 2) MV_BWD(2, G_, O_, B_, float, float, NT_, KQ_);         \
        else if (y16 && g16) MV_BWD(1, G_, O_, B_, _Float16, __bf16, NT_, KQ_);\
        else if (y16) MV_BWD(1, G_, O_, B_, _Float16, float, NT_, KQ_);        \
        else MV_BWD(1, G_, O_, B_, float, float, NT_, KQ_);                    \
    } while (0)
#define MV_BWD_SZ(G_, O_, B_)                                   \
    do {                                                        \
        if (NQ <= 160) MV_BWD_T(G_, O_, B_, 256, 5);            \
        else MV_BWD_T(G_, O_, B_, 512, 8);                      \
    } while (0)
#define MV_BWD_B(G_, O_)                                        \
    do {                                                        \
        if (base) MV_BWD_SZ(G_, O_, true);                      \
        else MV_BWD_SZ(G_, O_, false);                          \
    } while (0)
        if (gmode == 0 && omode == 0) MV_BWD_B(0, 0);
        else if (gmode == 0) MV_BWD_B(0, 1);
        else if (gmode == 1 && omode == 0) MV_BWD_B(1, 0);
        else if (gmode == 1) MV_BWD_B(1, 1);
        else if (omode == 0) MV_BWD_B(2, 0);
        else MV_BWD_B(2, 1);
#undef MV_BWD_B
#undef MV_BWD_SZ
#undef MV_BWD_T
#undef MV_BWD
        MV_LAUNCH_CHECK("cb_norm_bwd_q_kernel");
        return 0;
    }
    MV_REQUIRE(!y16 && !g16, E_SHAPE, "mdnvae_cb_norm_bwd: 16-bit planes need the LDS-staged kernel (planes too large or misaligned)");
    if (parts == 1) {
        if (PP <= 3 * 256) hipLaunchKernelGGL((cb_norm_bwd_kernel<1, 8, 3>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((cb_norm_bwd_kernel<1, 8, 9>), grid, dim3(256), 0, st, a);
    } else {
        if (PP <= 3 * 256) hipLaunchKernelGGL((cb_norm_bwd_kernel<2, 8, 3>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((cb_norm_bwd_kernel<2, 8, 9>), grid, dim3(256), 0, st, a);
    }
    MV_LAUNCH_CHECK("cb_norm_bwd_kernel");
    return 0;
}

size_t mdnvae_cb_wgrad_workspace_floats(int B, int Cout, int Cin, int planes, int parts) {
    int tco, tci;
    cb_wgrad_tiles(planes, parts, tco, tci);
    const int ntiles = ((Cout + tco - 1) / tco) * ((Cin + tci - 1) / tci);
    return (size_t)cb_wgrad_groups(B, ntiles) * 9 * Cout * Cin;
}

int mdnvae_cb_wgrad(const void* g, const void* act, float* workspace, float* grad_w, int B, int Cout, int Cin, int planes,
                    int PPg, int PPa, int kbeg, int kend, int pitch, int parts, float out_scale, long g_part_stride,
                    long act_part_stride, const unsigned* absmax_g, void* stream) {
    MV_REQUIRE(B >= 1, E_ARG, "mdnvae_cb_wgrad: empty batch");
    MV_REQUIRE(g && act && workspace && grad_w, E_ARG, "mdnvae_cb_wgrad: null pointer");
    MV_REQUIRE(planes == 1 || planes == 4, E_ARG, "mdnvae_cb_wgrad: planes = %d (1 or 4)", planes);
    MV_REQUIRE(parts == 1 || parts == 2, E_ARG, "mdnvae_cb_wgrad: parts = %d (1 or 2)", parts);
    MV_REQUIRE(parts == 1 || absmax_g, E_ARG, "mdnvae_cb_wgrad: two fp16 parts need the record the gradient was scaled by");
    MV_REQUIRE(Cout % 8 == 0 && Cin % 8 == 0, E_SHAPE, "mdnvae_cb_wgrad: channel counts must be multiples of 8");
    MV_REQUIRE(kbeg >= 0 && kbeg % 32 == 0 && kend % 32 == 0 && kend > kbeg && kend <= PPg, E_SHAPE,
               "mdnvae_cb_wgrad: position range [%d, %d) must be whole 32-position steps inside the %d-position plane", kbeg, kend, PPg);
    CbWgrad a{};
    a.g = static_cast<const char*>(g);
    a.act = static_cast<const char*>(act);
    a.partial = workspace;
    a.B = B; a.Cout = Cout; a.Cin = Cin; a.PPg = PPg; a.PPa = PPa; a.kbeg = kbeg; a.kend = kend;
    a.out_scale = out_scale;
    a.g_part_stride = g_part_stride; a.a_part_stride = act_part_stride; a.amax_g = parts == 2 ? absmax_g : nullptr;
    int smax;
    if (planes == 1) {
        a.smin = -2 * pitch - 2; smax = 0;
        for (int t = 0; t < 9; ++t) {
            a.tap_off[t] = (t / 3 - 2) * pitch + (t % 3 - 2) - a.smin;
            a.tap_plane[t] = 0;
        }
    } else {
        a.smin = 0; smax = pitch + 1;
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t % 3;
            a.tap_off[t] = (ky > 0) * pitch + (kx > 0);
            a.tap_plane[t] = (ky == 1 ? 0 : 1) * 2 + (kx == 1 ? 0 : 1);
        }
    }
    MV_REQUIRE(smax - a.smin <= 64, E_SHAPE, "mdnvae_cb_wgrad: pitch %d: the taps span %d positions (64 fit)", pitch, smax - a.smin);
    int tco, tci;
    cb_wgrad_tiles(planes, parts, tco, tci);
    const int ntiles = ((Cout + tco - 1) / tco) * ((Cin + tci - 1) / tci);
    const int Gz0 = cb_wgrad_groups(B, ntiles);
    a.spg = (B + Gz0 - 1) / Gz0;
    const int Gz = (B + a.spg - 1) / a.spg;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (parts == 2) {
        if (planes == 1) rc = cb_launch_wgrad<2, 1, 3, 2, 4, 64>(a, ntiles, Gz, st);
        else rc = cb_launch_wgrad<2, 4, 1, 8, 1, 64>(a, ntiles, Gz, st);
    } else if (planes == 1) rc = cb_launch_wgrad<1, 1, 3, 2, 4, 128>(a, ntiles, Gz, st);
    else rc = cb_launch_wgrad<1, 4, 3, 4, 2, 64>(a, ntiles, Gz, st);
    if (rc) return rc;
    const long total = 9L * Cout * Cin;
    hipLaunchKernelGGL(cb_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, workspace, Gz, Cout, Cin,
                       grad_w);
    MV_LAUNCH_CHECK("cb_wgrad_reduce_kernel");
    return 0;
}

}  // extern "C"
