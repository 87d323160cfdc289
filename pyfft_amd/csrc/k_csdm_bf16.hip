// k_csdm_bf16.hip -- cfg5's contraction G[k][i][j] = sum_g X_i[g,k] conj(X_j[g,k]) on the bf16 matrix cores with float32
// accuracy (generalises the reference's channel loop fft_analysis.py:387-393 / HeatPulse_Funcs.py:576-583).
//
// The fp32-input MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate, and k_csdm_fused is bound by it (80 % busy,
// 3.6 ms at 64 channels x 8191 frames x 2049 bins).  Here every float32 operand is split by truncation into three bf16 pieces
// x = h + m + l (8 significant bits each: the residual is <= 2^-24 |x|) and each real product x y is the eight piece
// products down to 2^-24: h h, m m, h m, m h, h l, l h, m l, l m.  The eight ride on the K axis of ONE MFMA: the A operand
// of a lane (8 K-slots) holds the pieces [h h | m m | h l | m l] of its element, the B operand [h m | h m | l h | l m], so
// that a v_mfma_f32_32x32x16_bf16 (32 cycles) does one real 32 x 32 outer-product update for TWO frames (one per half-wave) --
// 10 of them per frame pair and bin for the Hermitian 64 x 64 update against 10 fp32 MFMAs of 64 cycles: half the MFMA
// cycles, and every packed dword is written once into the register it is used from (a first version packed 6 products per
// frame into 3-dword groups spanning two frames: 118 of its 298 VALU instructions per tile were moves).
// Accumulation is the MFMA's float32.
//
// Layout: the STFT stage writes the spectra as Xs[pair][bin group of 8][channel slot of 64][8 bins][2 frames] (the two frames
// of a real-input transform side by side), so that the 8 bins of a workgroup are one 128-byte line per (channel, frame pair)
// and the 64 lines of one (pair, group) are 8 KiB contiguous (with the channels outermost every tile was 256 lines from 256
// DRAM rows and the kernel ran at the latency of that pattern: 2.9 us per 32 KiB tile).  One workgroup = 8 waves =
// 8 bins (wave w <-> bin k0 + w), 2 waves per SIMD with up to 256 VGPRs: 96 accumulators + the packed operands.  Lane l of a
// wave is channel l % 32 (and + 32) and frame l / 32 of each pair, as in k_csdm_fused.  Tiles of 4 frame pairs x 64 channels
// x 8 bins (32 KiB) go HBM -> registers -> LDS, double buffered, one barrier per tile.
#include "launch.h"
namespace sp {

#define CB_BINS 8
// CB_FORM: 8 = eight piece products per part, one operand per part and frame (10 MFMAs per pair); 6 = six products in
// 3-dword groups over two frames (8 MFMAs per pair)
#ifndef CB_FORM
#define CB_FORM 6
#endif
#ifndef CB_NT_LOADS
#define CB_NT_LOADS 1
#endif
#ifndef CB_INTERLEAVE
#define CB_INTERLEAVE 5          // VALU instructions per MFMA in the scheduling pipeline (0: leave it to hipcc)
#endif
#define CB_FP 4                                   // frame pairs per tile
#define CB_P 17                                   // LDS pitch (complex) of one (pair, channel) row of 8 bins x 2 frames
#define CB_TILE (CB_FP * 64 * CB_P)               // complex elements per buffer

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// x = h + m + l, each piece a bf16 = the top 16 bits of a float (truncation: residual < 2^-24 |x|).  The pieces are kept as
// the floats whose top halves they are (h: x itself, m: x - h, l: x - h - m) -- v_perm_b32 picks the top halves when packing.
struct Split3 {
    unsigned h, m, l;
};
__device__ __forceinline__ Split3 split3(float x) {
    Split3 s;
    s.h = __float_as_uint(x);
    const float r1 = x - __uint_as_float(s.h & 0xffff0000u);
    s.m = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(s.m & 0xffff0000u);
    s.l = __float_as_uint(r2);
    return s;
}
// dword holding the bf16 (top half) of a in its low half (K-slot 2j) and that of b in its high half (K-slot 2j + 1)
__device__ __forceinline__ unsigned pk(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// packed operands of one complex element for one frame: one MFMA operand (4 dwords = 8 K-slots) per part and role
struct Ops {
    bf16x8 ra, ia;        // A operands of the real / imaginary part:  [h h | m m | h l | m l]
    bf16x8 rb, ib;        // B operands:                               [h m | h m | l h | l m]
};
__device__ __forceinline__ bf16x8 op4(unsigned a, unsigned b, unsigned c, unsigned d) {
    u32x4 v = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ void make_part(float x, bf16x8 &a, bf16x8 &b) {
#if CB_ABLATE & 2
    const unsigned u = __float_as_uint(x);
    a = op4(u, u, u, u);
    b = a;
    return;
#endif
    const Split3 s = split3(x);
    // slots A|B: (h|h)(h|m)  (m|h)(m|m)  (h|l)(l|h)  (m|l)(l|m): no dword is common to A and B, so hipcc writes every
    // v_perm_b32 straight into its operand register (with [hm hm hl ml] | [hm mh lh lm] it built B as a copy of A: 12
    // instructions per part instead of 8)
    a = op4(pk(s.h, s.h), pk(s.m, s.m), pk(s.h, s.l), pk(s.m, s.l));
    b = op4(pk(s.h, s.m), pk(s.h, s.m), pk(s.l, s.h), pk(s.l, s.m));
}
__device__ __forceinline__ Ops make_ops(cf x) {
    Ops o;
    make_part(x.x, o.ra, o.rb);
    make_part(x.y, o.ia, o.ib);
    return o;
}
__device__ __forceinline__ bf16x8 neg8(bf16x8 a) {
    u32x4 v = __builtin_bit_cast(u32x4, a);
    v ^= 0x80008000u;
    return __builtin_bit_cast(bf16x8, v);
}
// CB_ABLATE (diagnostic builds, results wrong): 1 = no MFMAs (the operands are consumed by one add each), 2 = no operand
// preparation (the raw element bits as operands)
#ifndef CB_ABLATE
#define CB_ABLATE 0
#endif
#if CB_ABLATE & 1
#define CB_MFMA(acc, a, b)                                                                            \
    {                                                                                                 \
        const u32x4 ua_ = __builtin_bit_cast(u32x4, a), ub_ = __builtin_bit_cast(u32x4, b);           \
        acc[0] += __uint_as_float((ua_.x ^ ub_.y) + (ua_.z ^ ub_.w) + (ua_.y ^ ub_.x) + (ua_.w ^ ub_.z)); \
    }
#else
#define CB_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)
#endif

// FORM 4 (long records: csd_matrix_impl selects it from 1024 frame pairs on; SP_CSDM_SPLIT3=1 / SP_CSDM_SPLIT2=1 force a form): TWO pieces x ~ h + m (m rounded to nearest: 16 significant bits, residual <= 2^-16 |x|,
// zero mean) and the four products h h, h m, m h, m m: A = [h h | m m], B = [h m | h m] -- a part is 2 dwords, an element of one
// frame exactly one operand [re | im], so the Hermitian 64 x 64 update of a frame pair is 4 MFMAs + 1 for the two diagonal
// blocks' imaginary parts (two pairs share those): 10 per two pairs instead of 16.  The operands carry 16 bits instead of
// 24: the products' errors (rms 2^-17 / sqrt 3 each) average out over the frames but not for spectra that repeat exactly
// from frame to frame (measured there: 7e-7 of the peak against 3e-7, tools/split2_ab.py), hence only for long records where
// the float32 accumulation is the larger error anyway.
template <int FORM>
static __global__ __launch_bounds__(512) void k_csdm_bf16(const cf *__restrict__ Xs, int nch, int64_t npairs, int ld, int ngroups,
                                                          double *__restrict__ G, int64_t ps, int slices, int atomic) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *lds = reinterpret_cast<cf *>(smem_raw);
    const int unit = blockIdx.x;
    const int k0 = (unit / slices) * CB_BINS, zslice = unit % slices;
    const int64_t pbeg = (int64_t)zslice * ps, pend = pbeg + ps < npairs ? pbeg + ps : npairs;
    if (pbeg >= pend) return;
    const int ntiles = (int)((pend - pbeg + CB_FP - 1) / CB_FP);
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, half = lane >> 5, col = lane & 31;
    // staging: a tile row = (pair q, channel) = 8 bins x 2 frames = 128 bytes = 8 parts of 16 bytes; thread: part = t % 8,
    // channel = t / 8, the four pairs of the tile in turn
    const int part = t & 7, cl = t >> 3;
    const float keepc = cl < nch ? 1.f : 0.f;
    // Xs[pair][bin group][channel slot][8 bins][2 frames]: a tile row set (one pair, this group, 64 channels) is 8 KiB
    // contiguous and thread t takes its t-th 16 bytes
    const int64_t pstride = (int64_t)(ld / CB_BINS) * 512;              // float4 per pair
    const float4 *rowbase = reinterpret_cast<const float4 *>(Xs) + (int64_t)(k0 / CB_BINS) * 512 + (cl < nch ? t : part);
    cf *ldst = lds + cl * CB_P + 2 * part;
    float4 st[CB_FP];
    float keep[CB_FP];
    bool nomask = false;
    auto gfetch = [&](int64_t p0) __attribute__((always_inline)) {
        nomask = nch == 64 && p0 + CB_FP <= pend;
#pragma unroll
        for (int q = 0; q < CB_FP; ++q) {
            const int64_t p = p0 + q;
            const bool ok = p < pend;
#if CB_NT_LOADS
            {
                const sp_f4s r = __builtin_nontemporal_load(reinterpret_cast<const sp_f4s *>(rowbase + (ok ? p : pbeg) * pstride));
                st[q] = make_float4(r.x, r.y, r.z, r.w);       // read exactly once
            }
#else
            st[q] = rowbase[(ok ? p : pbeg) * pstride];        // clamped address; masked at lstore
#endif
            keep[q] = ok ? keepc : 0.f;
        }
    };
    auto lstore = [&](int bufsel) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < CB_FP; ++q) {
            cf *dst = ldst + bufsel * CB_TILE + (q * 64) * CB_P;
            if (nomask) {                                      // workgroup-uniform: 64 channels, whole tile inside the slice
                dst[0] = mk(st[q].x, st[q].y);
                dst[1] = mk(st[q].z, st[q].w);
            } else {
                dst[0] = mk(keep[q] * st[q].x, keep[q] * st[q].y);
                dst[1] = mk(keep[q] * st[q].z, keep[q] * st[q].w);
            }
        }
    };
    f32x16 accR[3], accI[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            accR[b][v] = 0.f;
            accI[b][v] = 0.f;
        }
    gfetch(pbeg);
    lstore(0);
    __syncthreads();
    // this lane's elements of a tile: (pair q, channel col / col + 32, bin `wave`, frame `half`)
    const cf *pa0 = lds + col * CB_P + 2 * wave + half;
    for (int it = 0; it < ntiles; ++it) {
        const bool more = it + 1 < ntiles;                    // workgroup-uniform
        if (more) gfetch(pbeg + (int64_t)(it + 1) * CB_FP);
        const cf *pa = pa0 + (it & 1) * CB_TILE;
        if constexpr (FORM == 4) {
            cf xe[CB_FP][2];
#pragma unroll
            for (int q = 0; q < CB_FP; ++q) {
                xe[q][0] = pa[(q * 64) * CB_P];
                xe[q][1] = pa[(q * 64 + 32) * CB_P];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pp = 0; pp < CB_FP / 2; ++pp) {
                unsigned A[2][2][2][2], B[2][2][2];              // [pair][channel][re/im]: A dwords (h h)(m m), B dword (h m)
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int part = 0; part < 2; ++part) {
                            const float x = part ? xe[2 * pp + f][c].y : xe[2 * pp + f][c].x;
                            const unsigned h = __float_as_uint(x);
                            const unsigned m = __float_as_uint(x - __uint_as_float(h & 0xffff0000u)) + 0x8000u;   // (the pack truncates)
                            A[f][c][part][0] = pk(h, h);
                            A[f][c][part][1] = pk(m, m);
                            B[f][c][part] = pk(h, m);
                        }
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const bf16x8 ar0 = op4(A[f][0][0][0], A[f][0][0][1], A[f][0][1][0], A[f][0][1][1]);
                    const bf16x8 ar1 = op4(A[f][1][0][0], A[f][1][0][1], A[f][1][1][0], A[f][1][1][1]);
                    const bf16x8 b0 = op4(B[f][0][0], B[f][0][0], B[f][0][1], B[f][0][1]);
                    const bf16x8 b1 = op4(B[f][1][0], B[f][1][0], B[f][1][1], B[f][1][1]);
                    const bf16x8 ai0 = op4(A[f][0][1][0], A[f][0][1][1], A[f][0][0][0] ^ 0x80008000u, A[f][0][0][1] ^ 0x80008000u);   // [i | -r]
                    CB_MFMA(accR[0], ar0, b0);
                    CB_MFMA(accR[1], ar0, b1);
                    CB_MFMA(accR[2], ar1, b1);
                    CB_MFMA(accI[1], ai0, b1);
                }
                // diagonal blocks: P = Xi Xr^T, the two pairs side by side on the K axis
                CB_MFMA(accI[0], op4(A[0][0][1][0], A[0][0][1][1], A[1][0][1][0], A[1][0][1][1]),
                        op4(B[0][0][0], B[0][0][0], B[1][0][0], B[1][0][0]));
                CB_MFMA(accI[2], op4(A[0][1][1][0], A[0][1][1][1], A[1][1][1][0], A[1][1][1][1]),
                        op4(B[0][1][0], B[0][1][0], B[1][1][0], B[1][1][0]));
            }
        } else {
#if CB_FORM == 6
        // six products per part ((h|h)(h|m) (m|h)(m|m) (h|l)(l|h): everything down to 2^-16 and the two largest 2^-24 terms) in
        // 3-dword groups; a lane's two frames of a pair of pairs make 12 dwords = three MFMA operands per block: 16 MFMAs per
        // two pairs instead of 20
        cf xe[CB_FP][2];
#pragma unroll
        for (int q = 0; q < CB_FP; ++q) {
            xe[q][0] = pa[(q * 64) * CB_P];
            xe[q][1] = pa[(q * 64 + 32) * CB_P];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pp = 0; pp < CB_FP / 2; ++pp) {
            Split3 s[2][2][2];                                   // [frame a/b][channel 0/1][re/im]
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    s[f][c][0] = split3(xe[2 * pp + f][c].x);
                    s[f][c][1] = split3(xe[2 * pp + f][c].y);
                }
            // group builders: A3 = [hh mm hl], B3 = [hm hm lh] of one part
#define A3_(S, j) ((j) == 0 ? pk(S.h, S.h) : (j) == 1 ? pk(S.m, S.m) : pk(S.h, S.l))
#define B3_(S, j) ((j) <= 1 ? pk(S.h, S.m) : pk(S.l, S.h))
            // dword d (0..11) of the 12-dword sequence [frame a: part P0, part P1][frame b: P0, P1] of channel c
#define SEQA_(c, d, P0, P1) A3_(s[(d) / 6][c][(((d) % 6) / 3) ? P1 : P0], (d) % 3)
#define SEQB_(c, d) B3_(s[(d) / 6][c][((d) % 6) / 3], (d) % 3)
#define QA_(c, q, P0, P1) op4(SEQA_(c, 4 * (q), P0, P1), SEQA_(c, 4 * (q) + 1, P0, P1), SEQA_(c, 4 * (q) + 2, P0, P1), SEQA_(c, 4 * (q) + 3, P0, P1))
#define QB_(c, q) op4(SEQB_(c, 4 * (q)), SEQB_(c, 4 * (q) + 1), SEQB_(c, 4 * (q) + 2), SEQB_(c, 4 * (q) + 3))
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const bf16x8 ar0 = QA_(0, q, 0, 1), ar1 = QA_(1, q, 0, 1), b0 = QB_(0, q), b1 = QB_(1, q);
                // imaginary part of block (0,1): A = [i part | -(r part)]: the sign bit of the r groups is flipped
                u32x4 ai = __builtin_bit_cast(u32x4, QA_(0, q, 1, 0));
                {
                    // dwords 4q..4q+3 of [a: i r][b: i r]: positions with ((d % 6) / 3) == 1 hold r groups
                    const unsigned m0 = (((4 * q) % 6) / 3) ? 0x80008000u : 0u, m1 = (((4 * q + 1) % 6) / 3) ? 0x80008000u : 0u;
                    const unsigned m2 = (((4 * q + 2) % 6) / 3) ? 0x80008000u : 0u, m3 = (((4 * q + 3) % 6) / 3) ? 0x80008000u : 0u;
                    ai.x ^= m0;
                    ai.y ^= m1;
                    ai.z ^= m2;
                    ai.w ^= m3;
                }
                const bf16x8 ai0 = __builtin_bit_cast(bf16x8, ai);
                CB_MFMA(accR[0], ar0, b0);
                CB_MFMA(accR[1], ar0, b1);
                CB_MFMA(accR[2], ar1, b1);
                CB_MFMA(accI[1], ai0, b1);
            }
            // diagonal blocks: P = Xi Xr^T: A = [a: A3(i)][b: A3(i)] 0 0, B = [a: B3(r)][b: B3(r)] 0 0
#define PA_(c, d) ((d) < 6 ? A3_(s[(d) / 3][c][1], (d) % 3) : 0u)
#define PB_(c, d) ((d) < 6 ? B3_(s[(d) / 3][c][0], (d) % 3) : 0u)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                CB_MFMA(accI[0], op4(PA_(0, 4 * q), PA_(0, 4 * q + 1), PA_(0, 4 * q + 2), PA_(0, 4 * q + 3)),
                        op4(PB_(0, 4 * q), PB_(0, 4 * q + 1), PB_(0, 4 * q + 2), PB_(0, 4 * q + 3)));
                CB_MFMA(accI[2], op4(PA_(1, 4 * q), PA_(1, 4 * q + 1), PA_(1, 4 * q + 2), PA_(1, 4 * q + 3)),
                        op4(PB_(1, 4 * q), PB_(1, 4 * q + 1), PB_(1, 4 * q + 2), PB_(1, 4 * q + 3)));
            }
#undef A3_
#undef B3_
#undef SEQA_
#undef SEQB_
#undef QA_
#undef QB_
#undef PA_
#undef PB_
        }
#else
        // all eight elements of the tile first (hipcc otherwise issued each ds_read right in front of its use: eight exposed
        // LDS round trips per tile, 30 % of the wave cycles at s_waitcnt)
        cf xe[CB_FP][2];
#pragma unroll
        for (int q = 0; q < CB_FP; ++q) {
            xe[q][0] = pa[(q * 64) * CB_P];
            xe[q][1] = pa[(q * 64 + 32) * CB_P];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < CB_FP; ++q) {
            // this lane's frame of pair q: channels col (rows / columns 0..31) and col + 32
            const Ops x0 = make_ops(xe[q][0]), x1 = make_ops(xe[q][1]);
            const bf16x8 nr0 = neg8(x0.ra);
            // Re G = Xr Xr^T + Xi Xi^T on the blocks (0,0), (0,1), (1,1); Im G = Xi Xr^T - Xr Xi^T on (0,1); on the diagonal
            // blocks only P = Xi Xr^T (Im G = P - P^T is taken once at the end)
            CB_MFMA(accR[0], x0.ra, x0.rb);
            CB_MFMA(accR[1], x0.ra, x1.rb);
            CB_MFMA(accR[2], x1.ra, x1.rb);
            CB_MFMA(accI[1], x0.ia, x1.rb);
            CB_MFMA(accI[0], x0.ia, x0.rb);
            CB_MFMA(accR[0], x0.ia, x0.ib);
            CB_MFMA(accR[1], x0.ia, x1.ib);
            CB_MFMA(accR[2], x1.ia, x1.ib);
            CB_MFMA(accI[1], nr0, x1.ib);
            CB_MFMA(accI[2], x1.ia, x1.rb);
        }
#endif
        }
#if CB_INTERLEAVE
        // ask the scheduler for MFMA / VALU interleaving: the operand preparation of the later K-slots runs in the shadow
        // of the earlier MFMAs (an MFMA holds the SIMD's issue for 8 of its 32 cycles)
#pragma unroll
        for (int i = 0; i < (FORM == 4 ? 5 : CB_FORM == 6 ? 8 : 10) * CB_FP; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x002, CB_INTERLEAVE, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
#endif
        if (more) lstore((it + 1) & 1);
        __syncthreads();
    }
    // register v of lane l is D[i = 8 (v/4) + 4 (l/32) + v%4][j = l%32]
    const int k = k0 + wave;
    float *tp = reinterpret_cast<float *>(lds) + wave * (32 * 33);       // this wave's 32 x 32 transpose image (pitch 33)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int bi = b == 2 ? 1 : 0, bj = b == 0 ? 0 : 1;
        if (b != 1) {
            // diagonal block: Im = P - P^T
            __syncthreads();
#pragma unroll
            for (int v = 0; v < 16; ++v) tp[(8 * (v / 4) + 4 * half + (v % 4)) * 33 + col] = accI[b][v];
            __syncthreads();
#pragma unroll
            for (int v = 0; v < 16; ++v) accI[b][v] -= tp[col * 33 + 8 * (v / 4) + 4 * half + (v % 4)];
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int i = 32 * bi + 8 * (v / 4) + 4 * half + (v % 4), j = 32 * bj + col;
            if (i < nch && j < nch) {
                double *p = G + (((int64_t)k * nch + i) * nch + j) * 2;
                if (atomic == 2) {                 // first contribution of an unsliced unit: store (G need not be zeroed)
                    p[0] = (double)accR[b][v];
                    p[1] = (double)accI[b][v];
                } else if (atomic) {
                    atomicAdd(p, (double)accR[b][v]);
                    atomicAdd(p + 1, (double)accI[b][v]);
                } else {
                    p[0] += (double)accR[b][v];
                    p[1] += (double)accI[b][v];
                }
            }
        }
    }
}

// tail bins (those that do not fill a group of 8: the Nyquist bin of every power-of-two nfft) for k_csdm_mfma:
// Xt[kk][g][c] = Xs[c][g / 2][kfirst + kk][g % 2], zero padded (kk < ntail <= 8)
static __global__ void k_csdm_gather_bins_pi(const cf *__restrict__ Xs, cf *__restrict__ Xt, int nch, int nchp, int64_t m, int64_t mp,
                                             int64_t npairs, int ld, int kfirst, int ntail) {
    const int64_t g = blockIdx.x;
    for (int e = threadIdx.x; e < ntail * nchp; e += blockDim.x) {
        const int kk = e / nchp, c = e % nchp;
        const bool ok = c < nch && g < m;
        const cf v = Xs[ok ? (((((g / 2) * (int64_t)(ld / CB_BINS) + kfirst / CB_BINS) * 64 + c) * CB_BINS + (kfirst % CB_BINS) + kk) * 2 + (g & 1)) : 0];
        Xt[((int64_t)kk * mp + g) * nchp + c] = ok ? v : mk(0.f, 0.f);
    }
}

// G[k] = (H[k] + conj H[n - k]) / 2, k = 0..n/2, on the blocks the contraction computes (j / 32 >= i / 32): H is the Hermitian
// contraction of the PACKED pair spectra Z = X_2q + i X_2q+1 over all n bins; the cross terms between the two frames of a pair
// cancel in the mirror combination (real signals: X[n-k] = conj X[k]), which is therefore taken once, on the sums.
// One workgroup per bin k.  st != null: the one-pass mean correction in the same sweep over G.
// The spectra were detrended by the estimates mu0_i; with d_i = mean_i - mu0_i (real), W = FFT(window), B_i = sum_g X_i,g:
//   sum_g (X_i - d_i W) conj(X_j - d_j W) = G_ij - d_j conj(W) B_i - d_i W conj(B_j) + M d_i d_j |W|^2
// st[ch] = state of k_op_finish<EXPORT>: B at [n .. 3n), the channel's plain sample sum at 5n + 3; nmean samples per channel
// init: G holds nothing yet (every chunk went through H): the result is STORED, scaled, and mirrored into the blocks below the
// diagonal blocks in the same sweep (k_csdm_finish / k_csdm_mirror and the zeroing of G are then not needed)
static __global__ __launch_bounds__(256) void k_csdm_fold(const double *__restrict__ H, double *__restrict__ G, int nch, int n,
                                                          const double *__restrict__ st, const cf *__restrict__ Wf,
                                                          const float *__restrict__ trend, int64_t nmean, int64_t M, double scale,
                                                          int init) {
    __shared__ double sd[64], sbr[64], sbi[64];
    const int64_t per = (int64_t)nch * nch;
    const int k = blockIdx.x;
    const int64_t km = (n - k) & (n - 1);
    double wr = 0.0, wi = 0.0;
    if (st) {
        const int64_t ss = (int64_t)5 * n + 8;
        if ((int)threadIdx.x < nch) {
            const int c = threadIdx.x;
            sd[c] = st[c * ss + 5 * n + 3] / (double)nmean - (double)trend[4 * c];
            sbr[c] = st[c * ss + n + 2 * k];
            sbi[c] = st[c * ss + n + 2 * k + 1];
        }
        wr = Wf[k].x;
        wi = Wf[k].y;
        __syncthreads();
    }
    const double mw = (double)M * (wr * wr + wi * wi);
    for (int ij = threadIdx.x; ij < (int)per; ij += 256) {
        const int j = ij % nch, i = ij / nch;
        if (j / 32 < i / 32) continue;
        const double2 a = reinterpret_cast<const double2 *>(H)[k * per + ij];
        const double2 b = reinterpret_cast<const double2 *>(H)[km * per + ij];
        double2 gv = init ? make_double2(0.0, 0.0) : reinterpret_cast<double2 *>(G)[k * per + ij];
        gv.x += 0.5 * (a.x + b.x);
        gv.y += 0.5 * (a.y - b.y);
        if (st) {
            const double di = sd[i], dj = sd[j];
            gv.x += -dj * (wr * sbr[i] + wi * sbi[i]) - di * (wr * sbr[j] + wi * sbi[j]) + mw * di * dj;
            gv.y += -dj * (wr * sbi[i] - wi * sbr[i]) - di * (wi * sbr[j] - wr * sbi[j]);
        }
        if (init) {
            gv.x *= scale;
            gv.y *= scale;
            if (j / 32 > i / 32) reinterpret_cast<double2 *>(G)[k * per + (int64_t)j * nch + i] = make_double2(gv.x, -gv.y);
        }
        reinterpret_cast<double2 *>(G)[k * per + ij] = gv;
    }
}
int launch_csdm_fold(LaunchCtx c, const double *H, double *G, int nch, int n, const double *st, const cf *Wf, const float *trend,
                     int64_t nmean, int64_t M, double scale, int init) {
    if (nch > 64) return -1;
    hipLaunchKernelGGL(k_csdm_fold, dim3(n / 2 + 1), dim3(256), 0, c.stream, H, G, nch, n, st, Wf, trend, nmean, M, scale, init);
    return 0;
}

// ---- one-pass channel means for the packed-spectra path ------------------------------------------------------------------
// Sl[ch][2j], Sl[ch][2j+1] = sum over the runs of the block sums spartial[ch][run][j] (float64, fixed order).  A workgroup takes
// 16 adjacent columns (one 128-byte line per run) in 16 run-slices, reduced through LDS by halving -- with one thread per column
// walking all runs, a call with few channels and hundreds of runs was one long latency chain on a handful of workgroups (0.3 ms)
static __global__ __launch_bounds__(256) void k_cm_blocksums(const cf *__restrict__ spartial, int runs, int hop, double *__restrict__ Sl) {
    __shared__ double sa[16][17], sb[16][17];
    const int lane = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + lane, ch = blockIdx.y;
    double a = 0.0, b = 0.0;
    if (j < hop) {
#pragma unroll 4
        for (int r = sl; r < runs; r += 16) {
            const cf v = spartial[((int64_t)ch * runs + r) * hop + j];
            a += (double)v.x;
            b += (double)v.y;
        }
    }
    sa[sl][lane] = a;
    sb[sl][lane] = b;
    __syncthreads();
    for (int o = 8; o > 0; o >>= 1) {
        if (sl < o) {
            sa[sl][lane] += sa[sl + o][lane];
            sb[sl][lane] += sb[sl + o][lane];
        }
        __syncthreads();
    }
    if (sl == 0 && j < hop) {
        Sl[(int64_t)ch * 2 * hop + 2 * j] = sa[0][lane];
        Sl[(int64_t)ch * 2 * hop + 2 * j + 1] = sb[0][lane];
    }
}
int launch_cm_blocksums(LaunchCtx c, const cf *spartial, int nch, int runs, int hop, double *Sl) {
    hipLaunchKernelGGL(k_cm_blocksums, dim3((hop + 15) / 16, nch), dim3(256), 0, c.stream, spartial, runs, hop, Sl);
    return 0;
}
// Xs: pair-interleaved spectra [nch][npairs][ld][2] of m frames (the second frame of an odd last pair is zero)
// (Xs may also hold PACKED pair spectra with `m` pairs as frames and nb = nfft bins: see k_csdm_fold)
// init: G holds nothing yet -- an unsliced launch without tail bins stores its sums, otherwise G is zeroed here first
int launch_csdm_bf16(LaunchCtx c, const cf *Xs, cf *Xt_tail, int nch, int64_t m, int nb, double *G, int ld, int two_pieces, int init) {
    if (nch > 64 || ld < nb || (ld % CB_BINS) != 0) return -1;
    const int64_t npairs = (m + 1) / 2;
    const int nchp = 64;
    const int ngroups = (nb % CB_BINS) == 0 ? nb / CB_BINS : (nb - 1) / CB_BINS;      // (a whole number of groups: no tail)
    if (ngroups > 0) {
        // frame-pair slices when there are fewer bin groups than CUs; sliced units add atomically
        int slices = (c.ncu + ngroups - 1) / ngroups;
        const int max_slices = (int)((npairs + 63) / 64);
        if (slices > max_slices) slices = max_slices;
        if (slices < 1) slices = 1;
        int64_t ps = (npairs + slices - 1) / slices;
        ps = (ps + CB_FP - 1) / CB_FP * CB_FP;
        slices = (int)((npairs + ps - 1) / ps);
        const size_t lds = 2 * sizeof(cf) * CB_TILE;                              // 68 KiB
        static bool attr_done = false;
        if (!attr_done) {
            if (hipFuncSetAttribute((const void *)k_csdm_bf16<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
                hipFuncSetAttribute((const void *)k_csdm_bf16<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return -1;
            attr_done = true;
        }
        int mode = slices > 1 ? 1 : 0;
        if (init) {
            if (slices == 1 && nb == CB_BINS * ngroups) mode = 2;
            else if (hipMemsetAsync(G, 0, sizeof(double) * 2 * (size_t)nb * (size_t)nch * (size_t)nch, c.stream) != hipSuccess) return -1;
        }
        if (two_pieces)
            hipLaunchKernelGGL(k_csdm_bf16<4>, dim3(ngroups * slices), dim3(512), lds, c.stream, Xs, nch, npairs, ld, ngroups, G, ps,
                               slices, mode);
        else
            hipLaunchKernelGGL(k_csdm_bf16<0>, dim3(ngroups * slices), dim3(512), lds, c.stream, Xs, nch, npairs, ld, ngroups, G, ps,
                               slices, mode);
    }
    const int kfirst = CB_BINS * ngroups, ntail = nb - kfirst;
    if (ntail > 0) {
        const int64_t mp = (m + 31) / 32 * 32;
        hipLaunchKernelGGL(k_csdm_gather_bins_pi, dim3((unsigned)mp), dim3(256), 0, c.stream, Xs, Xt_tail, nch, nchp, m, mp, npairs, ld,
                           kfirst, ntail);
        if (launch_csdm_mfma(c, Xt_tail, nch, nchp, mp, ntail, G + (int64_t)kfirst * nch * nch * 2)) return -1;
    }
    return 0;
}

}   // namespace sp
