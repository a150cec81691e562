// pmx_nwsg16.hip -- fast kernel for global (nw) and semi-global (sg, any free-end set) alignment,
// score + end positions, many independent pairs per launch.  gfx950 only.
//
// Dispatch names `nw_striped_{16,sat,32,64}`, `sg[_q?][_d?]_striped_{16,sat,32,64}`
// (/root/reference/src/aligner/mod.rs:289-331) when no stats / table / trace output is asked for.
//
// Same machine mapping as pmx_sw16.hip: G lanes per slot, two pairs per slot in the int16 halves of
// every register, R query rows per lane in VGPRs, the reference streamed one column per step with a
// one-step skew between neighbouring lanes (DPP), per-pair query profile in LDS.  Differences:
//
//   * Values are biased by 16384 so that negative scores are representable: the exact window is
//     true value in [-15360, 15359] (bit patterns [1024, 31743], where v_pk_maximum3_f16 is an exact
//     integer max3).  The host proves from lengths, gap penalties and matrix extremes that no cell can
//     leave the window; otherwise the general 32-bit kernel is used.
//   * The query is aligned to the BOTTOM of the G*R-row strip set, so its last row is always the last
//     register of the last lane.  The P = G*R - qlen rows above it are *virtual rows*, the G-1 pad
//     symbols in front of the reference are *virtual columns*.  Their profile scores are chosen so
//     that the ordinary recurrences reproduce the boundary conditions exactly:
//        penalised side  : score -inf, the F (resp. E) chain carries -(open + k*extend)
//        free side (sg)  : score 0, the diagonal carries 0
//        virtual x virtual : score 0 (an all-zero corner)
//     (needs open >= extend, which the reference asks for: src/aligner/mod.rs:139-153).
//     No lane ever needs a special case: lane 0's "row above" is the constant pair (H = 0, F = -inf).
//   * Results are captured, not tracked: nw reads the last row at column rlen-1; sg keeps the first
//     maximum of the last row (reference end free) and of the last column (query end free; last column
//     wins only if strictly greater -- oracle/pmx_oracle.c states the rule).
#include "pmx_common.h"
#include "pmx_switches.h"
#include <cstdlib>

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2us __attribute__((ext_vector_type(2)));

#define PK(x)  __builtin_bit_cast(v2s, (int)(x))
#define I32(x) __builtin_bit_cast(int, (x))

#ifndef PMX_QSTAGE
#define PMX_QSTAGE 8                   // steps of trace records the shared-profile sweep gathers in LDS per flush (4 or 8)
#endif
#define NB 16384                       // bias
#define NB2 ((NB << 16) | NB)
#define NEGS ((short)-32768)           // "-inf" score: sets the sign bit of the biased sum

__device__ __forceinline__ v2s n_addw(v2s a, v2s b)       // wrapping add (v_pk_add_u16)
{
    return __builtin_bit_cast(v2s, __builtin_bit_cast(v2us, a) + __builtin_bit_cast(v2us, b));
}
__device__ __forceinline__ v2s n_subus(v2s a, v2s b)      // v_pk_sub_u16 clamp
{
    return __builtin_bit_cast(v2s, __builtin_elementwise_sub_sat(__builtin_bit_cast(v2us, a), __builtin_bit_cast(v2us, b)));
}
__device__ __forceinline__ v2s n_max3(v2s a, v2s b, v2s c)
{
    int r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(I32(a)), "v"(I32(b)), "v"(I32(c)));
    return PK(r);
}
typedef _Float16 n_v2h __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s n_max3f(v2s a, v2s b, v2s c)     // same instruction, as a builtin (no inline-asm wait states)
{
    const n_v2h r = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(n_v2h, a), __builtin_bit_cast(n_v2h, b)),
                                                  __builtin_bit_cast(n_v2h, c));
    return __builtin_bit_cast(v2s, r);
}
__device__ __forceinline__ v2s n_min3f(v2s a, v2s b, v2s c)     // v_pk_minimum3_f16: exact integer min3 on the same patterns
{
    const n_v2h r = __builtin_elementwise_minimum(__builtin_elementwise_minimum(__builtin_bit_cast(n_v2h, a), __builtin_bit_cast(n_v2h, b)),
                                                  __builtin_bit_cast(n_v2h, c));
    return __builtin_bit_cast(v2s, r);
}
__device__ __forceinline__ int n_bfi(int m, int a, int b)
{
    int r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int s_bfi(int m /* scalar constant */, int a, int b)
{
    int r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(m), "v"(a), "v"(b));
    return r;
}
// per-half masks (0xFFFF / 0) on values below 32768
__device__ __forceinline__ int m_lt(v2s a, v2s b) { const v2s sh = {15, 15}; return I32((a - b) >> sh); }      // a < b
__device__ __forceinline__ int m_eq(v2s a, v2s b)
{
    const v2us one = {1, 1};
    const v2us x = __builtin_bit_cast(v2us, I32(a) ^ I32(b));
    return I32(__builtin_bit_cast(v2s, __builtin_elementwise_min(x, one) - one));
}
__device__ __forceinline__ int m_ult(v2s a, v2s b)        // unsigned a < b (a may be a "negative" column index)
{
    const v2us one = {1, 1}, zero = {0, 0};
    const v2us d = __builtin_elementwise_sub_sat(__builtin_bit_cast(v2us, b), __builtin_bit_cast(v2us, a));
    return I32(__builtin_bit_cast(v2s, zero - __builtin_elementwise_min(d, one)));
}

// IL (G == 8): two groups interleaved in a DPP row of 16 lanes (lane = 2 g + (slot & 1) + 16 (slot >> 1)): row_shr:2
// moves every group up by one lane and the row's first two lanes keep `neutral` -- no select (see pmx_sw16.hip).
template <int G, bool IL = false>
__device__ __forceinline__ int n_shift_up(int x, int neutral, int g)
{
    if (IL) return __builtin_amdgcn_update_dpp(neutral, x, 0x112 /*row_shr:2*/, 0xF, 0xF, false);
    if (G <= 16) {
        int r = __builtin_amdgcn_update_dpp(neutral, x, 0x111 /*row_shr:1*/, 0xF, 0xF, false);
        if (G < 16) r = (g == 0) ? neutral : r;
        return r;
    } else {
        int r = __builtin_amdgcn_update_dpp(neutral, x, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
        if (G < 64) r = (g == 0) ? neutral : r;
        return r;
    }
}

template <int G, int R>
__global__ __launch_bounds__(64)
void pmx_nwsg16_kernel(const uint8_t *__restrict__ qbuf, const int64_t *__restrict__ qoff,
                       const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff,
                       long long n, const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap,
                       int msize, int open, int ext, int RP, int q_shared,
                       int col_pen /* H(i,-1) penalised */, int row_pen /* H(-1,j) penalised */,
                       int s1_end /* query end free */, int s2_end /* reference end free */,
                       const unsigned *__restrict__ perm,
                     pmx_record_t *__restrict__ out)
{
    static_assert(R % 2 == 0, "rows are stored two per dword");
    constexpr int QP = G * R;
    constexpr int QP2 = QP / 2;
    constexpr int SLOTS = 64 / G;
    constexpr int NP = 2 * SLOTS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int lane = threadIdx.x;
    const int g = lane % G;
    const int slot = lane / G;
    const int MS1 = msize + 1;                      // + the pair's own pad-symbol row
    const int PROF_STRIDE = MS1 * QP * 2;

    // LDS carve: [prof NP][rsym NP*RP][mat][map 256][pair table NP*4 ints]
    int16_t *prof = reinterpret_cast<int16_t *>(lds);
    unsigned char *rsym = lds + NP * PROF_STRIDE;
    int16_t *mat = reinterpret_cast<int16_t *>(rsym + NP * RP);
    unsigned char *map = reinterpret_cast<unsigned char *>(mat + msize * msize);
    long long *ptab = reinterpret_cast<long long *>(map + 256 + ((8 - ((msize * msize * 2) & 7)) & 7));   // per pair: q offset, qlen, r offset, rlen, pair index

    const long long pair0 = (long long)blockIdx.x * NP;

    for (int i = lane; i < msize * msize; i += 64) mat[i] = gmat[i];
    for (int i = lane; i < 256; i += 64) map[i] = gmap[i];
    if (lane < NP) {
        long long pos = pair0 + lane; if (pos >= n) pos = n - 1;
        const long long pi = perm ? (long long)perm[pos] : pos;
        const long long qb = q_shared ? 0 : qoff[pi], rb = roff[pi];
        ptab[5 * lane + 0] = qb;
        ptab[5 * lane + 1] = q_shared ? q_shared : (qoff[pi + 1] - qb);
        ptab[5 * lane + 2] = rb;
        ptab[5 * lane + 3] = roff[pi + 1] - rb;
        ptab[5 * lane + 4] = (pair0 + lane < n) ? pi : -1;
    }
    __syncthreads();
    const uint8_t *qbase = qbuf;
    const uint8_t *rbase = rbuf;

    // ---- reference symbols, G-1 virtual columns in front, pad behind ---------------------
    int max_rlen = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) max_rlen = max(max_rlen, (int)ptab[5 * p + 3]);
    constexpr int UB = 8;
    for (int item0 = 0; item0 < NP * RP; item0 += 64 * UB) {
        unsigned char raw[UB]; bool ok[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int item = item0 + u * 64 + lane;
            const int p = min(item / RP, NP - 1), j = item - p * RP;
            const int jj = j - (G - 1);
            ok[u] = item < NP * RP && jj >= 0 && jj < (int)ptab[5 * p + 3];
            raw[u] = ok[u] ? rbase[ptab[5 * p + 2] + jj] : (unsigned char)0;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int item = item0 + u * 64 + lane;
            if (item < NP * RP) rsym[item] = ok[u] ? map[raw[u]] : (unsigned char)msize;
        }
    }

    // ---- extended profiles: P virtual rows on top of the qlen real rows --------------------
    const int vrow_score = row_pen ? NEGS : 0;     // virtual row  x real symbol
    const int vcol_score = col_pen ? NEGS : 0;     // real row     x pad symbol
    constexpr int QITEMS = (NP * QP2 + 63) / 64;
    constexpr int QB = QITEMS < 5 ? QITEMS : 5;
    for (int it0 = 0; it0 < QITEMS; it0 += QB) {
        unsigned char r0[QB], r1[QB]; bool v0[QB], v1[QB];
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const int item = (it0 + u) * 64 + lane;
            const int p = min(item / QP2, NP - 1), rp = item - p * QP2;
            const int P = QP - (int)ptab[5 * p + 1];
            const uint8_t *qp = qbase + ptab[5 * p + 0];
            v0[u] = item < NP * QP2 && 2 * rp >= P; v1[u] = item < NP * QP2 && 2 * rp + 1 >= P;
            r0[u] = v0[u] ? qp[2 * rp - P] : (unsigned char)0;
            r1[u] = v1[u] ? qp[2 * rp + 1 - P] : (unsigned char)0;
        }
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const int item = (it0 + u) * 64 + lane;
            if (it0 + u < QITEMS && item < NP * QP2) {
                const int p = item / QP2, rp = item - p * QP2;
                const int q0 = v0[u] ? map[r0[u]] : -1;
                const int q1 = v1[u] ? map[r1[u]] : -1;
                int *pp = reinterpret_cast<int *>(prof) + p * (PROF_STRIDE / 4) + rp;
                for (int sym = 0; sym < msize; ++sym) {
                    const int s0 = (q0 < 0) ? vrow_score : mat[q0 * msize + sym];
                    const int s1 = (q1 < 0) ? vrow_score : mat[q1 * msize + sym];
                    pp[sym * QP2] = (s0 & 0xFFFF) | (s1 << 16);
                }
                const int p0 = (q0 < 0) ? 0 : vcol_score, p1 = (q1 < 0) ? 0 : vcol_score;
                pp[msize * QP2] = (p0 & 0xFFFF) | (p1 << 16);
            }
        }
    }
    __syncthreads();

    // ---- per-lane state ------------------------------------------------------------------
    const int pA = 2 * slot, pB = 2 * slot + 1;
    const unsigned char *profA = lds + pA * PROF_STRIDE + g * (R * 2);
    const unsigned char *profB = lds + pB * PROF_STRIDE + g * (R * 2);
    const unsigned char *rsA = rsym + pA * RP + (G - 1) - g;
    const unsigned char *rsB = rsym + pB * RP + (G - 1) - g;
    const int SYMSTRIDE = QP * 2;

    const int PvA = QP - (int)ptab[5 * pA + 1], PvB = QP - (int)ptab[5 * pB + 1];   // virtual rows per half
    const int rlA = (int)ptab[5 * pA + 3], rlB = (int)ptab[5 * pB + 3];
    const v2s vOpen = PK((open & 0xFFFF) | (open << 16));
    const v2s vExt = PK((ext & 0xFFFF) | (ext << 16));
    const v2us one2 = {1, 1};

    // H(row, -1): 0 for virtual rows and for a free query begin, -(open + i*ext) otherwise
    auto left_h = [&](int erow, int P) -> int {
        const int i = erow - P;
        return NB + ((i >= 0 && col_pen) ? -(open + i * ext) : 0);
    };
    auto pack2 = [](int a, int b) -> int { return (a & 0xFFFF) | (b << 16); };

    v2s HA[R], HB[R], E[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int hA = left_h(g * R + k, PvA), hB = left_h(g * R + k, PvB);
        HA[k] = PK(pack2(hA, hB)); HB[k] = HA[k];
        E[k] = PK(pack2(hA - open, hB - open));
    }
    int Hout = I32(HA[R - 1]);
    // F flowing below the strip at virtual columns: -(open + i*ext) for a penalised real row below, else -open
    auto below_f = [&](int erow, int P) -> int {
        const int i = erow - P;
        return NB + ((i >= 0 && col_pen) ? -(open + i * ext) : -open);
    };
    int Fout = pack2(below_f((g + 1) * R, PvA), below_f((g + 1) * R, PvB));
    v2s diag0 = (g == 0) ? PK(NB2) : PK(pack2(left_h(g * R - 1, PvA), left_h(g * R - 1, PvB)));

    const v2s rl1 = PK(pack2(rlA - 1, rlB - 1)), rlv = PK(pack2(rlA, rlB));
    int jj = ((-g) & 0xFFFF) * 0x00010001;
    int res = NB2;                         // nw / sg without free ends: H(qlen-1, rlen-1)
    v2s bestrow = PK(0); int bestrowj = 0; // sg, reference end free: first max of the last row
    v2s bestcol = PK(0); int bestcoli = 0; // sg, query end free: first max of the last column (extended row index)

    auto load_scores = [&](int symA, int symB, int (&wa)[R / 2], int (&wb)[R / 2]) {
        const int *sa = reinterpret_cast<const int *>(profA + symA * SYMSTRIDE);
        const int *sb = reinterpret_cast<const int *>(profB + symB * SYMSTRIDE);
#pragma unroll
        for (int k = 0; k < R / 2; ++k) { wa[k] = sa[k]; wb[k] = sb[k]; }
    };
    auto step = [&](const v2s (&Hold)[R], v2s (&Hnew)[R], const int (&wa)[R / 2], const int (&wb)[R / 2]) {
        const int Hin = n_shift_up<G>(Hout, NB2, g);     // row above lane 0: H = 0
        v2s F = PK(n_shift_up<G>(Fout, 0, g));           //                   F = -inf
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const v2s s = PK(__builtin_amdgcn_perm(wb[k / 2], wa[k / 2], (k & 1) ? 0x07060302 : 0x05040100));
            const v2s d = (k == 0) ? diag0 : Hold[k - 1];
            const v2s Tt = n_addw(d, s);
            const v2s H = n_max3(Tt, E[k], F);
            const v2s Ho = n_subus(H, vOpen);
            E[k] = n_max3(n_subus(E[k], vExt), Ho, Ho);
            F = n_max3(n_subus(F, vExt), Ho, Ho);
            Hnew[k] = H;
        }
        diag0 = PK(Hin);
        Hout = I32(Hnew[R - 1]);
        Fout = I32(F);

        // ---- captures ----
        const v2s jv = PK(jj);
        const int mLast = m_eq(jv, rl1);                  // this lane is at column rlen-1
        res = n_bfi(mLast, Hout, res);
        if (s2_end) {
            const int imp = m_lt(bestrow, PK(Hout)) & m_ult(jv, rlv);
            bestrow = PK(n_bfi(imp, Hout, I32(bestrow)));
            bestrowj = n_bfi(imp, jj, bestrowj);
        }
        if (s1_end && __builtin_amdgcn_ballot_w64(mLast != 0) != 0) {
            // last column of this strip: maximum over the REAL rows, smallest row first
            const v2s Pv = PK(pack2(PvA, PvB));
            v2s cm = PK(0); int krow = 0;
            v2s vals[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int er = g * R + k;
                const int mreal = ~m_lt(PK(pack2(er, er)), Pv);
                vals[k] = PK(I32(Hnew[k]) & mreal);
                cm = n_max3(cm, vals[k], vals[k]);
            }
#pragma unroll
            for (int k = R - 1; k >= 0; --k) {
                const int er = g * R + k;
                krow = n_bfi(m_eq(vals[k], cm), pack2(er, er), krow);
            }
            const int imp = m_lt(bestcol, cm) & mLast;
            bestcol = PK(n_bfi(imp, I32(cm), I32(bestcol)));
            bestcoli = n_bfi(imp, krow, bestcoli);
        }
        jj = I32(__builtin_bit_cast(v2s, __builtin_bit_cast(v2us, jj) + one2));
    };

    const int T = (max_rlen + G - 1 + 1) & ~1;
    int w0a[R / 2], w0b[R / 2], w1a[R / 2], w1b[R / 2];
    load_scores(rsA[0], rsB[0], w0a, w0b);
    int nsA = rsA[1], nsB = rsB[1];
    for (int t = 0; t < T; t += 2) {
        load_scores(nsA, nsB, w1a, w1b);
        nsA = rsA[t + 2]; nsB = rsB[t + 2];
        __builtin_amdgcn_sched_barrier(0);
        step(HA, HB, w0a, w0b);
        __builtin_amdgcn_sched_barrier(0);
        load_scores(nsA, nsB, w0a, w0b);
        nsA = rsA[t + 3]; nsB = rsB[t + 3];
        __builtin_amdgcn_sched_barrier(0);
        step(HB, HA, w1a, w1b);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- combine -----------------------------------------------------------------------
    // last-column candidates: reduce (value desc, extended row asc) over the slot's lanes
    unsigned keyA = ((unsigned)(I32(bestcol) & 0xFFFF) << 16) | (0xFFFFu - (unsigned)(bestcoli & 0xFFFF));
    unsigned keyB = ((unsigned)((unsigned)I32(bestcol) >> 16) << 16) | (0xFFFFu - ((unsigned)bestcoli >> 16));
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) {
        const unsigned oa = __shfl_xor(keyA, off, 64), ob = __shfl_xor(keyB, off, 64);
        keyA = oa > keyA ? oa : keyA;
        keyB = ob > keyB ? ob : keyB;
    }
    // the last lane of the slot owns the last row: corner value and last-row maximum
    const int lastlane = slot * G + G - 1;
    const int resL = __shfl(res, lastlane, 64);
    const int browL = __shfl(I32(bestrow), lastlane, 64), browjL = __shfl(bestrowj, lastlane, 64);
    if (g == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long pi = ptab[5 * (2 * slot + h) + 4];
            if (pi >= 0) {
                const int ql = (int)ptab[5 * (2 * slot + h) + 1], rl = (int)ptab[5 * (2 * slot + h) + 3];
                const int P = QP - ql;
                const int corner = ((h ? ((unsigned)resL >> 16) : (resL & 0xFFFF))) - NB;
                pmx_record_t rec;
                rec.flags = 0;
                if (!s1_end && !s2_end) { rec.score = corner; rec.end_query = ql - 1; rec.end_ref = rl - 1; }
                else {
                    int best = -2147483647 - 1, ei = 0, ej = 0;
                    if (s2_end) {
                        best = (int)(h ? ((unsigned)browL >> 16) : (browL & 0xFFFF)) - NB;
                        ei = ql - 1; ej = (int)(h ? ((unsigned)browjL >> 16) : (browjL & 0xFFFF));
                    }
                    if (s1_end) {
                        const unsigned key = h ? keyB : keyA;
                        const int cv = (int)(key >> 16) - NB;
                        if (cv > best) { best = cv; ei = (int)(0xFFFFu - (key & 0xFFFFu)) - P; ej = rl - 1; }
                    }
                    rec.score = best; rec.end_query = ei; rec.end_ref = ej;
                }
                out[pi] = rec;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Second generation (same mapping, same results): the arithmetic of pmx_sw16.hip's fastest variant.
//   * column-skewed values: a value of column j is kept as value + nb + (j + G) * ext, so
//     E(j+1) = max(E(j) - ext, H(j) - open) needs no subtract; F keeps one more +ext so that the single
//     X = H - (open - ext) serves E, F and the strip (the next column's diagonal source);
//   * the profile carries score + open (one byte, >= 0), so add and subtract never carry or borrow
//     between the int16 halves and run as 32-bit VOP2;
//   * no "-inf": a penalised virtual row / column scores -open (profile byte 0), which can never beat
//     the E / F chain that carries -(open + k ext) there (open >= ext); the row above lane 0 is the
//     closed form of that chain (a constant in the skewed domain) or, for a free reference begin, 0;
//   * the bias nb is chosen by the host from the proven value range (room for the skew growth).
//   * TR: the same sweep also writes the 4-bit traceback cells pmx_trace16.hip's walk reads (bit 3 ND: T < H,
//     bit 2 NDL: F < H, bit 1 EO: E of the next column opened, bit 0 FO: F of the next row opened).  Each
//     decision is the sign of one packed difference, pushed into a packed plane (16 bits = 4 rows per
//     half): 3 instructions per decision and 2 cells.  Per lane and step one 16-byte store, coalesced over the
//     wave: [pair A rows 0-7, A rows 8-15, B rows 0-7, B rows 8-15], row 0 in the top nibble.  
//     TRB (the host proves max score + 2 open <= 250): every one of the four differences lies in [-256, 255], so its sign fills
//     bits 15..8 of its half and a decision is inserted at ITS bit position with one v_bfi_b32 -- no shift: two rows' eight
//     decisions chain through one register (4 v_pk_sub_i16 + 3.5 v_bfi_b32 per row), one v_perm_b32 per four rows packs the top bytes.
//   * FETCH (long references): reference symbols are not staged in LDS, each lane fetches its next symbols from HBM
//     two steps ahead (see pmx_sw16q.hip); the staged copies of four 5-kaa references would halve the occupancy.
//   * PT (alphabets of <= 4 letters + wildcard, as pmx_sw16.hip's VAR 6): NO LDS profile -- the v_perm that widened the profile
//     bytes is the lookup itself (table = the 4 scores of this step's reference symbol per pair, one dword each, from a
//     (msize + 1)-entry LDS table; selector = the lane's query letters, one VGPR per row).  LDS per wave drops from ~15 KB to
//     ~3 KB, so the VGPR count alone decides the occupancy.  A selector byte can only pick a table byte or the constants 0x00 /
//     0xFF, which cannot express the virtual rows of the bottom-aligned layout for a free reference begin: the PT form is
//     TOP-aligned -- the row above lane 0 is the closed form anyway, padding rows below the query score -open and feed nothing
//     -- and reads the query's last row (corner, last-row maximum) from register (qlen - 1) % R of lane (qlen - 1) / R with one
//     indexed move.  That index must be wave-uniform, and a query wildcard has no selector: a block whose pairs differ in query
//     length or hold a letter beyond the first four marks itself in `blockflag` and leaves; the launcher then runs the LDS-
//     profile form over exactly the marked blocks (`only_flagged`), and the walk reads the alignment of a block's rows from the
//     same flags.
//   * ROWX (row offset, round 3): every value of effective row er is stored + er * ext, on top of the column skew that does the same
//     for E along a row -- the vertical gap then needs no subtraction per row either: F(er) = max(F(er - 1), X(er - 1)) as it
//     stands.  The diagonal step crosses one row: + ext in every score byte; boundary values, hand-offs and captures carry their
//     row's offset; the decisions compare values of one row.  One instruction per row (of 7; 15.5 with the trace).  Off only for
//     width 8's range tracking, which compares H across rows in every step.
template <int G, int R, bool TR, bool FETCH = false, bool TRB = false, bool PT = false, bool ROWX = true>
__global__ __launch_bounds__(64, (PT && !TR && R <= 20) ? 4 : (PT && TR) ? 3 : 1)      // (PT + TR: the staged records' LDS allows 2.75 waves per SIMD)
void pmx_nwsg16v_kernel(const uint8_t *__restrict__ qbuf, const int64_t *__restrict__ qoff,
                        const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff,
                        long long n, const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap,
                        int msize, int open, int ext, int RP, int q_shared,
                        int col_pen, int row_pen, int s1_end, int s2_end, int nb,
                        const unsigned *__restrict__ perm,
                        pmx_record_t *__restrict__ out, uint32_t *__restrict__ tbuf, int Tmax,
                        int track8 /* width 8: report whether some H (boundaries included) leaves [-128, 127] */,
                        int *__restrict__ blockflag /* PT: 1 = this block is left to the LDS-profile form */,
                        const int *__restrict__ only_flagged /* LDS-profile form behind a PT launch: only the blocks marked there */)
{
    static_assert(!TR || R == 16, "trace: four packed planes of 4 rows");
    if (!PT && only_flagged && only_flagged[blockIdx.x] == 0) return;
    constexpr int RS = (R + 3) / 4 * 4;      // profile bytes reserved per lane (whole dwords)
    constexpr int QP = G * R;                // logical rows (query bottom-aligned in them; PT: top-aligned)
    constexpr int QPS = G * RS;              // profile bytes per (pair, symbol)
    constexpr int SLOTS = 64 / G;
    constexpr int NP = 2 * SLOTS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int lane = threadIdx.x;
    constexpr bool IL = G == 8 && !TR;          // interleaved 8-lane groups; the trace layout keeps plain groups
    const int g = IL ? (lane % 16) / 2 : lane % G;
    const int slot = IL ? (lane / 16) * 2 + (lane & 1) : lane / G;
    const int MS1 = msize + 1;                      // + the pad-symbol row
    const int PROF_STRIDE = PT ? 0 : MS1 * QPS;

    unsigned char *rsym = lds + NP * PROF_STRIDE;
    int16_t *mat = reinterpret_cast<int16_t *>(rsym + (FETCH ? 0 : NP * RP));
    unsigned char *map = reinterpret_cast<unsigned char *>(mat + msize * msize);
    long long *ptab = reinterpret_cast<long long *>(map + 256 + ((8 - ((msize * msize * 2) & 7)) & 7));
    int *tabs = reinterpret_cast<int *>(ptab + 5 * NP);      // PT: per reference symbol the 4 query-letter scores (+ open); entry msize = the pad symbol
    constexpr int QS = (G * R + 3) / 4 * 4;
    unsigned char *qsym = reinterpret_cast<unsigned char *>(tabs + 40);   // PT: mapped query letters, QS bytes per pair (0xFF below the query)
    // PT + TR: the LDS the profiles no longer take holds eight steps of every lane's trace records, which then leave as ONE aligned
    // 128-byte line per lane stream, eight lanes per line (pmx_nwsg16q_kernel's flush).  Two steps per 32-byte sector straight from
    // registers (the LDS-profile form) leave every line of every resident lane stream partly written in the L2 for eight steps --
    // more lines than the L2s hold: that form's sweep is bound by its write path, not by its instructions (a fourth wave per SIMD
    // bought it nothing).
    constexpr bool STG = PT && TR;
    constexpr int TSTG = 8, TSTR = TSTG * 4 + 1;             // steps per flush; dwords per lane (odd: a step's stores of the lanes fall into different banks)
    uint32_t *tstage = reinterpret_cast<uint32_t *>(qsym + NP * QS + 16) + (STG ? lane * TSTR : 0);

    const long long pair0 = (long long)blockIdx.x * NP;
    for (int i = lane; i < msize * msize; i += 64) mat[i] = gmat[i];
    for (int i = lane; i < 256; i += 64) map[i] = gmap[i];
    if (lane < NP) {
        long long pos = pair0 + lane; if (pos >= n) pos = n - 1;
        const long long pi = perm ? (long long)perm[pos] : pos;
        const long long qb = q_shared ? 0 : qoff[pi], rb = roff[pi];
        ptab[5 * lane + 0] = qb;
        ptab[5 * lane + 1] = q_shared ? q_shared : (qoff[pi + 1] - qb);
        ptab[5 * lane + 2] = rb;
        ptab[5 * lane + 3] = roff[pi + 1] - rb;
        ptab[5 * lane + 4] = (pair0 + lane < n) ? pi : -1;
    }
    __syncthreads();
    constexpr int UB = NP < 8 ? NP : 8;
    int qlu = 0;                                             // PT: the block's common query length
    if (PT) {
        // the block's eligibility: one query length, no letter beyond the first four (decided before anything else is staged)
        qlu = (int)ptab[1];
        bool ok = true;
#pragma unroll
        for (int p = 1; p < NP; ++p) ok = ok && (int)ptab[5 * p + 1] == qlu;
        int wild = 0;
        for (int p0 = 0; p0 < NP; p0 += UB) {
            for (int j0 = 0; j0 < QS; j0 += 64) {
                const int j = j0 + lane;
                unsigned char raw[UB]; bool in[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    in[u] = j < (int)ptab[5 * (p0 + u) + 1];
                    raw[u] = in[u] ? qbuf[ptab[5 * (p0 + u) + 0] + j] : (unsigned char)0;
                }
                if (j < QS) {
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int c = in[u] ? (int)map[raw[u]] : 0xFF;
                        wild |= (c >= 4 && c != 0xFF) ? 1 : 0;
                        qsym[(p0 + u) * QS + j] = (unsigned char)c;
                    }
                }
            }
        }
        ok = ok && __builtin_amdgcn_ballot_w64(wild != 0) == 0 && qlu <= QP;
        if (!TR && track8) {
            // width 8 on this (untracked) form: only blocks whose pairs ALL saturate by their boundary alone -- a penalised boundary
            // column or row that runs past -128: reads of >= 63 bp under 5 / 2 -- whose flag therefore needs no look at the table;
            // every other block is left to the form that tracks the range of H
#pragma unroll
            for (int p = 0; p < NP; ++p)
                ok = ok && ((col_pen && open + ((int)ptab[5 * p + 1] - 1) * ext > 128) || (row_pen && open + ((int)ptab[5 * p + 3] - 1) * ext > 128));
        }
        if (lane == 0) blockflag[blockIdx.x] = ok ? 0 : 1;
        if (!ok) return;                                     // (wave-uniform)
        if (lane <= msize) {
            int v = 0;
            if (lane < msize) { for (int k = 0; k < 4 && k < msize; ++k) v |= ((mat[k * msize + lane] + open + (ROWX ? ext : 0)) & 0xFF) << (8 * k); }
            else v = ((col_pen ? 0 : open) + (ROWX ? ext : 0)) * 0x01010101;      // real row x virtual / padding column
            tabs[lane] = v;
        }
    }

    // ---- reference symbols (pairs in batches of UB, lanes over positions: no division, UB loads in flight)
    int max_rlen = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) max_rlen = max(max_rlen, (int)ptab[5 * p + 3]);
    for (int p0 = 0; p0 < (FETCH ? 0 : NP); p0 += UB) {
        for (int j0 = 0; j0 < RP; j0 += 64) {
            const int j = j0 + lane, jr = j - (G - 1);
            unsigned char raw[UB]; bool ok[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int p = p0 + u;
                ok[u] = jr >= 0 && jr < (int)ptab[5 * p + 3];
                raw[u] = ok[u] ? rbuf[ptab[5 * p + 2] + jr] : (unsigned char)0;
            }
            if (j < RP) {
#pragma unroll
                for (int u = 0; u < UB; ++u) rsym[(p0 + u) * RP + j] = ok[u] ? map[raw[u]] : (unsigned char)msize;
            }
        }
    }

    // ---- byte profiles: logical row er = l * R + k sits at byte l * RS + k; P virtual rows on top
    const int rx = ROWX ? ext : 0;                 // the row offset per row
    const int vrow_b = (row_pen ? 0 : open) + rx;  // virtual row x real symbol
    const int vcol_b = (col_pen ? 0 : open) + rx;  // real row    x pad symbol
    for (int p0 = 0; p0 < (PT ? 0 : NP); p0 += UB) {
        for (int e0 = 0; e0 < QP; e0 += 64) {
            const int er = e0 + lane;
            unsigned char raw[UB]; bool real[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int p = p0 + u;
                const int P = QP - (int)ptab[5 * p + 1];
                real[u] = er < QP && er >= P;
                raw[u] = real[u] ? qbuf[ptab[5 * p + 0] + er - P] : (unsigned char)0;
            }
            if (er < QP) {
                const int pos = (er / R) * RS + er % R;
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    unsigned char *pp = lds + (p0 + u) * PROF_STRIDE + pos;
                    if (real[u]) {
                        const int q = map[raw[u]];
                        for (int sym = 0; sym < msize; ++sym) pp[sym * QPS] = (unsigned char)(mat[q * msize + sym] + open + rx);
                        pp[msize * QPS] = (unsigned char)vcol_b;
                    } else {
                        for (int sym = 0; sym < msize; ++sym) pp[sym * QPS] = (unsigned char)vrow_b;
                        pp[msize * QPS] = (unsigned char)(open + rx);   // virtual x virtual: score 0
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- per-lane state ------------------------------------------------------------------
    const int pA = 2 * slot, pB = 2 * slot + 1;
    const unsigned char *profA = lds + pA * PROF_STRIDE + g * RS;
    const unsigned char *profB = lds + pB * PROF_STRIDE + g * RS;
    const unsigned char *rsA = rsym + pA * RP + (G - 1) - g;
    const unsigned char *rsB = rsym + pB * RP + (G - 1) - g;

    const int PvA = PT ? 0 : QP - (int)ptab[5 * pA + 1], PvB = PT ? 0 : QP - (int)ptab[5 * pB + 1];
    const int rlA = (int)ptab[5 * pA + 3], rlB = (int)ptab[5 * pB + 3];
    // PT: per-row selectors (byte 0: pair A's letter -> table A = v_perm source bytes 0..3, byte 2: 4 + pair B's letter -> table B =
    // bytes 4..7, bytes 1 and 3: the constant 0; rows below the query select the constant 0 = score -open, and feed nothing)
    int sel[PT ? R : 1];
    const int gs = PT ? (qlu - 1) / R : G - 1, ks = PT ? (qlu - 1) % R : R - 1;       // lane and register of the query's last row
    if (PT) {
        const unsigned char *qa = qsym + pA * QS + g * R, *qb_ = qa + QS;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int ca = qa[k], cb = qb_[k];
            sel[k] = (ca < 4 ? ca : 0x0C) | 0x0C00 | ((cb < 4 ? 4 + cb : 0x0C) << 16) | 0x0C000000;
        }
    }
    auto pack2 = [](int a, int b) -> int { return (a & 0xFFFF) | (b << 16); };
    const int vExt = pack2(ext, ext), vC = pack2(open - ext, open - ext);
    const v2us one2 = {1, 1};
    const int base = nb + (G - g) * ext - open;    // X-form of a true 0 in this lane's column j0 - 1 (and E~ of column j0)

    auto left_h = [&](int erow, int P) -> int {    // true H(row, virtual column)
        int i = erow - P;
        if (PT) i = min(i, qlu - 1);               // (padding rows below the query: any value inside the proven window)
        return (i >= 0 && col_pen) ? -(open + i * ext) : 0;
    };
    auto below_f = [&](int erow, int P) -> int {   // true F flowing into row erow at a virtual column
        int i = erow - P;
        if (PT) i = min(i, qlu - 1);
        return (i >= 0 && col_pen) ? -(open + i * ext) : -open;
    };

    int HA[R], HB[R], E[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int ro = (g * R + k) * rx;             // (row offset of effective row g R + k)
        HA[k] = pack2(base + left_h(g * R + k, PvA) + ro, base + left_h(g * R + k, PvB) + ro);
        HB[k] = HA[k]; E[k] = HA[k];
    }
    int Hout = HA[R - 1];
    const int roF = ((g + 1) * R - 1) * rx, roD = (g * R - 1) * rx;       // the running F of row er carries (er - 1) rx; the diagonal comes from row g R - 1
    int Fout = pack2(base + open + below_f((g + 1) * R, PvA) + roF, base + open + below_f((g + 1) * R, PvB) + roF);
    int diag0 = (g == 0) ? pack2(base + roD, base + roD) : pack2(base + left_h(g * R - 1, PvA) + roD, base + left_h(g * R - 1, PvB) + roD);
    // row above lane 0: penalised -> H(-1, j) = -(open + j ext): a constant in the skewed X-form; free -> 0: grows by ext per column
    int topX = row_pen ? pack2(nb + (G + 1) * ext - 2 * open - rx, nb + (G + 1) * ext - 2 * open - rx)
                       : pack2(nb + (G + 1) * ext - open - rx, nb + (G + 1) * ext - open - rx);     // lane 0, column 0 (row -1)
    const int topStep = row_pen ? 0 : vExt;
    const int roL = ((PT ? qlu : QP) - 1) * rx;    // row offset of the query's last row
    // ROWX: the bias nb no longer covers the decline along the gaps (the offsets cancel it in what is STORED), but the free-end captures
    // compare values with skew and offset taken off again: they carry a bias of their own, cb >= -(lowest true H) (scores + open >= 0,
    // so -min <= open), taken off with nb at the end.  The host proves nb + cb + highest H < 2^15.
    const int cb = ROWX ? 4 * open + (QP + max_rlen + 2) * ext : 0;
    int skewX = pack2((G - g + 1) * ext - open + roL - cb, (G - g + 1) * ext - open + roL - cb);   // X-form minus nb minus cb minus true value (last row), this lane's column j0; += ext

    const v2s rl1 = PK(pack2(rlA - 1, rlB - 1)), rlv = PK(pack2(rlA, rlB));
    int jj = ((-g) & 0xFFFF) * 0x00010001;
    int res = 0;                           // X-form (skew of column rlen-1) of H(qlen-1, rlen-1)
    v2s bestrow = PK(0); int bestrowj = 0; // sg, reference end free: first max of the last row, UNSKEWED X-form (nb + H)
    v2s bestcol = PK(0); int bestcoli = 0; // sg, query end free: first max of the last column (skew of column rlen-1)
    // width 8 (`nw_*_8`, `sg*_8`): the reference's narrowest -- and on a CPU fastest -- width reports saturation when some H of the
    // table leaves the int8 range (oracle/pmx_oracle.c, src/alignment/mod.rs:436-440).  The int16 lanes compute the same table:
    // a running maximum and minimum of the columns' H over the cells inside the table (virtual rows hold boundary values, which
    // count; virtual and padding columns are masked) travel with the column skew; the boundary row / column join in closed form.
    v2s runmax = PK(HA[0] + vExt), runmin = runmax;    // a boundary value of this lane's row (HA: form of column j0 - 1), in the form of its first column j0

    auto load_scores = [&](int symA, int symB, int (&wa)[RS / 4], int (&wb)[RS / 4]) {
        if (PT) { wa[0] = tabs[symA]; wb[0] = tabs[symB]; return; }
        const int *sa = reinterpret_cast<const int *>(profA + symA * QPS);
        const int *sb = reinterpret_cast<const int *>(profB + symB * QPS);
#pragma unroll
        for (int k = 0; k < RS / 4; ++k) { wa[k] = sa[k]; wb[k] = sb[k]; }
    };
    // trace records of 16 bytes per lane and step, lane-major: every lane's steps are contiguous, so what the walk reads along a row
    // or a diagonal sits in one cache line (consecutive stores of a lane fill its 128-byte lines in L2)
    const size_t t_ss = 4;
    uint32_t *tw = TR ? tbuf + ((size_t)blockIdx.x * Tmax) * 256 + (size_t)lane * Tmax * 4 : nullptr;
    uint4 wprev = {0u, 0u, 0u, 0u};
    const v2s two2 = {2, 2}, sh15 = {15, 15};
    auto push = [&](v2s &pl, int a, int b) {         // pl = 2 * pl + (a < b), per half
        const v2s bit = PK(I32(__builtin_bit_cast(v2us, PK(a) - PK(b)) >> __builtin_bit_cast(v2us, sh15)));
        int r;                                        // asm: left alone the compiler regroups the pushes into more instructions
        asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(I32(pl)), "v"(I32(two2)), "v"(I32(bit)));
        pl = PK(r);
    };
    auto step = [&](const int (&Hold)[R], int (&Hnew)[R], const int (&wa)[RS / 4], const int (&wb)[RS / 4], int t) {
        const int Hin = n_shift_up<G, IL>(Hout, topX, g);
        int F = n_shift_up<G, IL>(Fout, topX, g);            // F^ into row 0 = X of the row above (see the header)
        int Tpre[R];
        v2s plane[TR ? R / 4 : 1];
        int tacc = 0, tprev = 0;
        if (TR) {
#pragma unroll
            for (int x = 0; x < R / 4; ++x) plane[x] = PK(0);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int s = PT ? __builtin_amdgcn_perm(wb[0], wa[0], (unsigned)sel[k])
                             : __builtin_amdgcn_perm(wb[k / 4], wa[k / 4], 0x0C000C00u | (unsigned)(k & 3) | ((4u + (unsigned)(k & 3)) << 16));
            Tpre[k] = ((k == 0) ? diag0 : Hold[k - 1]) + s;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int Fe = ROWX ? F : F - vExt;
            const int H = I32(n_max3f(PK(Tpre[k]), PK(E[k]), PK(Fe)));
            const int X = H - vC;
            if (TR && !TRB) {
                push(plane[k / 4], Tpre[k], H);      // ND
                push(plane[k / 4], Fe, H);           // NDL
                push(plane[k / 4], E[k], X);         // EO: E(j) - ext < H - open in the skewed forms
                push(plane[k / 4], Fe, X);           // FO: F - ext < H - open
            }
            if (TR && TRB) {
                // sign of (a - b) at bit `pos` of each half; differences are within [-256, 255] (see the header)
                const int dND = I32(PK(Tpre[k]) - PK(H)), dNDL = I32(PK(Fe) - PK(H));
                const int dEO = I32(PK(E[k]) - PK(X)), dFO = I32(PK(Fe) - PK(X));
                if ((k & 1) == 0) {
                    tacc = dND;                                                   // bit 15 (the rest is replaced below)
                    tacc = s_bfi(0x40004000, dNDL, tacc);
                    tacc = s_bfi(0x20002000, dEO, tacc);
                    tacc = s_bfi(0x10001000, dFO, tacc);
                } else {
                    tacc = s_bfi(0x08000800, dND, tacc);
                    tacc = s_bfi(0x04000400, dNDL, tacc);
                    tacc = s_bfi(0x02000200, dEO, tacc);
                    tacc = s_bfi(0x01000100, dFO, tacc);
                    if ((k & 3) == 1) tprev = tacc;                              // rows 4x, 4x+1
                    else plane[k / 4] = PK(__builtin_amdgcn_perm(tprev, tacc, 0x07030501));   // + rows 4x+2, 4x+3: top bytes of the halves
                }
            }
            E[k] = I32(n_max3f(PK(E[k]), PK(X), PK(X)));
            F = I32(n_max3f(PK(Fe), PK(X), PK(X)));
            Hnew[k] = X;
        }
        if (TR) {
            uint4 w;
            w.x = __builtin_amdgcn_perm(I32(plane[0]), I32(plane[1]), 0x00010405);   // A: bytes = row pairs (0,1) (2,3) (4,5) (6,7), even row in the high nibble
            w.y = __builtin_amdgcn_perm(I32(plane[2]), I32(plane[3]), 0x00010405);
            w.z = __builtin_amdgcn_perm(I32(plane[0]), I32(plane[1]), 0x02030607);   // B
            w.w = __builtin_amdgcn_perm(I32(plane[2]), I32(plane[3]), 0x02030607);
            // two steps leave together: 32 contiguous bytes per lane = one whole sector (single 16-byte stores of thousands of
            // resident lanes reach HBM as partly written sectors: PMC WRITE_SIZE was 1.7x the bytes stored; four steps per group
            // cost more registers than they save: 26.6 ms against 25.6 ms per cfg-4 step)
            if (STG) { uint32_t *dst = tstage + (t & (TSTG - 1)) * 4; dst[0] = w.x; dst[1] = w.y; dst[2] = w.z; dst[3] = w.w; }
            else if (t & 1) { uint4 *dst = reinterpret_cast<uint4 *>(tw + (size_t)(t - 1) * t_ss); dst[0] = wprev; dst[1] = w; }
            else wprev = w;
        }
        diag0 = Hin;
        Hout = Hnew[R - 1];
        Fout = F;

        // ---- captures ----
        const v2s jv = PK(jj);
        if (!TR && !ROWX && track8) {                              // (wave-uniform: the other widths pay one scalar branch per step)
            v2s cmx = PK(Hnew[0]), cmn = PK(Hnew[0]);
#pragma unroll
            for (int k = 1; k < R; k += 2) {
                const int k2 = k + 1 < R ? k + 1 : k;
                cmx = n_max3f(cmx, PK(Hnew[k]), PK(Hnew[k2]));
                cmn = n_min3f(cmn, PK(Hnew[k]), PK(Hnew[k2]));
            }
            const int inside = m_ult(jv, rlv);            // this lane's column lies in [0, rlen)
            runmax = PK(n_bfi(inside, I32(n_max3f(runmax, cmx, cmx)), I32(runmax)));
            runmin = PK(n_bfi(inside, I32(n_min3f(runmin, cmn, cmn)), I32(runmin)));
            runmax = PK(I32(runmax) + vExt); runmin = PK(I32(runmin) + vExt);      // into the next column's skew
        }
        const int mLast = m_eq(jv, rl1);                  // this lane is at column rlen-1
        // H of the query's last row, where this lane holds it.  PT: a wave-uniform register index -- one indexed move for strips of
        // up to 16 registers; longer strips (the allocator would move them to scratch for it) pick the register by a select chain in
        // the rare step that captures the corner, and leave batches that need the last row in every step to the LDS-profile form
        int Hlast = Hout;
        if (PT && R <= 16) Hlast = Hnew[ks];
        if (PT && R > 16) {
            if (__builtin_amdgcn_ballot_w64(mLast != 0) != 0) {
                int hl = Hnew[0];
#pragma unroll
                for (int k = 1; k < R; ++k) hl = (ks == k) ? Hnew[k] : hl;
                res = n_bfi(mLast, hl, res);
            }
        } else res = n_bfi(mLast, Hlast, res);
        if (s2_end) {
            if (PT && R > 16) {                           // the last row in every step from a long strip: a select chain (one per row; the
                Hlast = Hnew[0];                          //  perm-table form's occupancy is worth several times that)
#pragma unroll
                for (int k = 1; k < R; ++k) Hlast = (ks == k) ? Hnew[k] : Hlast;
            }
            const v2s cand = PK(I32(__builtin_bit_cast(v2s, __builtin_bit_cast(v2us, Hlast) - __builtin_bit_cast(v2us, skewX))));   // nb + true H
            const int imp = m_lt(bestrow, cand) & m_ult(jv, rlv);
            bestrow = PK(n_bfi(imp, I32(cand), I32(bestrow)));
            bestrowj = n_bfi(imp, jj, bestrowj);
        }
        if (s1_end && __builtin_amdgcn_ballot_w64(mLast != 0) != 0) {
            const v2s Pv = PK(pack2(PvA, PvB));
            v2s cm = PK(0); int krow = 0;
            v2s vals[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int er = g * R + k;
                const int mreal = PT ? (er < qlu ? -1 : 0) : ~m_lt(PK(pack2(er, er)), Pv);
                vals[k] = PK((Hnew[k] + pack2((QP - er) * rx, (QP - er) * rx)) & mreal);       // rows compare in the form of the LAST row's offset (+ (QP - er) rx: inside the proven window, which holds the rows' offsets)
                cm = n_max3f(cm, vals[k], vals[k]);
            }
#pragma unroll
            for (int k = R - 1; k >= 0; --k) {
                const int er = g * R + k;
                krow = n_bfi(m_eq(vals[k], cm), pack2(er, er), krow);
            }
            const int imp = m_lt(bestcol, cm) & mLast;
            bestcol = PK(n_bfi(imp, I32(cm), I32(bestcol)));
            bestcoli = n_bfi(imp, krow, bestcoli);
        }
        jj = I32(__builtin_bit_cast(v2s, __builtin_bit_cast(v2us, jj) + one2));
        skewX = I32(__builtin_bit_cast(v2s, __builtin_bit_cast(v2us, skewX) + __builtin_bit_cast(v2us, vExt)));   // halves may be negative: per-half add
        topX += topStep;
    };

    const int T = (max_rlen + G - 1 + 1) & ~1;
    int w0a[RS / 4], w0b[RS / 4], w1a[RS / 4], w1b[RS / 4];
    const uint8_t *refA = rbuf + ptab[5 * pA + 2], *refB = rbuf + ptab[5 * pB + 2];
    auto fetch = [&](int x, int &ra, int &rb) {      // FETCH: raw byte of step x (column x - g), -1 outside the reference
        const int col = x - g;
        ra = (col >= 0 && col < rlA) ? (int)refA[col] : -1;
        rb = (col >= 0 && col < rlB) ? (int)refB[col] : -1;
    };
    auto sym_of = [&](int raw) -> int { return raw < 0 ? msize : (int)map[raw]; };
    int m2a = 0, m2b = 0, m3a = 0, m3b = 0, nsA, nsB;
    if (FETCH) {
        int r0a, r0b, r1a, r1b;
        fetch(0, r0a, r0b); fetch(1, r1a, r1b); fetch(2, m2a, m2b); fetch(3, m3a, m3b);
        load_scores(sym_of(r0a), sym_of(r0b), w0a, w0b);
        nsA = sym_of(r1a); nsB = sym_of(r1b);
    } else {
        load_scores(rsA[0], rsB[0], w0a, w0b);
        nsA = rsA[1]; nsB = rsB[1];
    }
    for (int t = 0; t < T; t += 2) {
        load_scores(nsA, nsB, w1a, w1b);
        if (FETCH) { nsA = sym_of(m2a); nsB = sym_of(m2b); fetch(t + 4, m2a, m2b); }
        else { nsA = rsA[t + 2]; nsB = rsB[t + 2]; }
        __builtin_amdgcn_sched_barrier(0);
        step(HA, HB, w0a, w0b, t);
        __builtin_amdgcn_sched_barrier(0);
        load_scores(nsA, nsB, w0a, w0b);
        if (FETCH) { nsA = sym_of(m3a); nsB = sym_of(m3b); fetch(t + 5, m3a, m3b); }
        else { nsA = rsA[t + 3]; nsB = rsB[t + 3]; }
        __builtin_amdgcn_sched_barrier(0);
        step(HB, HA, w1a, w1b, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (STG && ((t & (TSTG - 2)) == (TSTG - 2) || t + 2 >= T)) {             // eight steps gathered (or the sweep ends)
            // item w of the flush = 16-byte piece w % 8 of lane stream w / 8: every store instruction writes whole 128-byte lines
            const uint32_t *wstage = tstage - lane * TSTR;                          // the wave's staging area
            uint32_t *wtw = tw - (size_t)lane * Tmax * 4;                           // the wave's lane stream 0
#pragma unroll
            for (int x = 0; x < TSTG; ++x) {
                const int w = x * 64 + lane, stream = w >> 3, piece = w & 7;
                const uint32_t *src = wstage + stream * TSTR + piece * 4;
                uint4 v; v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
                *reinterpret_cast<uint4 *>(wtw + ((size_t)stream * Tmax + (size_t)(t & ~(TSTG - 1))) * 4 + piece * 4) = v;
            }
        }
    }

    // ---- combine (all captured values -> true scores) ---------------------------------------
    unsigned keyA = ((unsigned)(I32(bestcol) & 0xFFFF) << 16) | (0xFFFFu - (unsigned)(bestcoli & 0xFFFF));
    unsigned keyB = ((unsigned)((unsigned)I32(bestcol) >> 16) << 16) | (0xFFFFu - ((unsigned)bestcoli >> 16));
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) {
        const int lo = IL ? 2 * off : off;
        const unsigned oa = __shfl_xor(keyA, lo, 64), ob = __shfl_xor(keyB, lo, 64);
        keyA = oa > keyA ? oa : keyA;
        keyB = ob > keyB ? ob : keyB;
    }
    const int lastlane = IL ? 2 * gs + (slot & 1) + 16 * (slot >> 1) : slot * G + gs;       // the lane that holds the query's last row
    const int resL = __shfl(res, lastlane, 64);
    const int browL = __shfl(I32(bestrow), lastlane, 64), browjL = __shfl(bestrowj, lastlane, 64);
    int hiA = 0, hiB = 0, loA = 0, loB = 0;                // width 8: true extremes of H over the group's cells (0 = H(-1, -1) counts)
    if (!TR && !ROWX && track8) {
        const int un = nb + (T - g + G) * ext - open + ext;                      // X-form of a true 0 in the form the running values ended in
        hiA = max(0, (I32(runmax) & 0xFFFF) - un); hiB = max(0, (int)((unsigned)I32(runmax) >> 16) - un);
        loA = min(0, (I32(runmin) & 0xFFFF) - un); loB = min(0, (int)((unsigned)I32(runmin) >> 16) - un);
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
            const int lo = IL ? 2 * off : off;
            hiA = max(hiA, __shfl_xor(hiA, lo, 64)); hiB = max(hiB, __shfl_xor(hiB, lo, 64));
            loA = min(loA, __shfl_xor(loA, lo, 64)); loB = min(loB, __shfl_xor(loB, lo, 64));
        }
    }
    if (g == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long pi = ptab[5 * (2 * slot + h) + 4];
            if (pi >= 0) {
                const int ql = (int)ptab[5 * (2 * slot + h) + 1], rl = (int)ptab[5 * (2 * slot + h) + 3];
                const int P = PT ? 0 : QP - ql;
                const int unsk = nb + (rl - 1 + G) * ext - open + ext;          // X-form of a true 0 at column rlen-1
                const int corner = (int)(h ? ((unsigned)resL >> 16) : (resL & 0xFFFF)) - unsk - roL;
                pmx_record_t rec;
                rec.flags = 0;
                if (!TR && PT && track8) rec.flags = PMX_FLAG_SATURATED;       // (this form only keeps blocks that saturate by their boundary)
                if (!TR && !ROWX && track8) {
                    int lo = h ? loB : loA;
                    const int hi = h ? hiB : hiA;
                    if (col_pen) lo = min(lo, -(open + (ql - 1) * ext));        // H(i, -1), H(-1, j): the boundary column and row
                    if (row_pen) lo = min(lo, -(open + (rl - 1) * ext));
                    if (hi > 127 || lo < -128) rec.flags = PMX_FLAG_SATURATED;
                }
                if (!s1_end && !s2_end) { rec.score = corner; rec.end_query = ql - 1; rec.end_ref = rl - 1; }
                else {
                    int best = -2147483647 - 1, ei = 0, ej = 0;
                    if (s2_end) {
                        best = (int)(h ? ((unsigned)browL >> 16) : (browL & 0xFFFF)) - nb - cb;
                        ei = ql - 1; ej = (int)(h ? ((unsigned)browjL >> 16) : (browjL & 0xFFFF));
                    }
                    if (s1_end) {
                        const unsigned key = h ? keyB : keyA;
                        const int cv = (int)(key >> 16) - unsk - QP * rx;
                        if (cv > best) { best = cv; ei = (int)(0xFFFFu - (key & 0xFFFFu)) - P; ej = rl - 1; }
                    }
                    rec.score = best; rec.end_query = ei; rec.end_ref = ej;
                }
                out[pi] = rec;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Shared query (profile arm: `nw_*_profile_*`, `sg*_profile_*`, /root/reference/src/aligner/mod.rs:431-450): the
// second-generation arithmetic with ONE profile per 4-wave workgroup (virtual rows included: they are the same for
// every pair) and reference symbols fetched from HBM two steps ahead instead of being staged in LDS -- what
// pmx_sw16q.hip does for local alignment.  A protein profile plus four staged 5-kaa references cost 58 KB per
// wave otherwise.
// TR: the same sweep also writes the packed traceback records (R / 2 bytes per pair, lane and step; see pmx_walkp.hip) with the
// one-instruction decision merge of pmx_nwsg16v_kernel's TRB form -- what BASELINE config 3 (statistics of a reused profile
// against long references) runs on: the statistics are counted along the path afterwards.
// ENDS = false: the instance for GLOBAL alignment -- no free-end captures compiled in.  The rare capture branches sit on top of the
// sweep's ~165 live registers: <16,20,TR> takes 178 VGPRs with them (two waves per SIMD; forced to 168 it spills and is 10 % slower)
// and 149 without (three waves per SIMD) -- BASELINE config 3 is global.
template <int G, int R, int WAVES, bool TR = false, bool ENDS = true>
__global__ __launch_bounds__(64 * WAVES)
void pmx_nwsg16q_kernel(const uint8_t *__restrict__ qbuf, int qlen,
                        const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff,
                        long long n, const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap,
                        int msize, int open, int ext,
                        int col_pen, int row_pen, int s1_end_arg, int s2_end_arg, int nb,
                        const unsigned *__restrict__ perm,
                        pmx_record_t *__restrict__ out, uint32_t *__restrict__ tbuf = nullptr, int Tmax = 0)
{
    static_assert(!TR || R == 10 || R == 16 || R == 19 || R == 20, "trace record layouts");
    const int s1_end = ENDS ? s1_end_arg : 0, s2_end = ENDS ? s2_end_arg : 0;
    constexpr int TD = (R + 3) / 4;               // dwords per trace record (R / 2 bytes per pair, two pairs)
    constexpr int RS = (R + 3) / 4 * 4;
    constexpr int QP = G * R;
    constexpr int QPS = G * RS;
    constexpr int NPW = 2 * (64 / G);
    constexpr int NP = NPW * WAVES;
    constexpr int NT = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane % G, slotw = lane / G;
    const int pA = wave * NPW + 2 * slotw, pB = pA + 1;
    const int MS1 = msize + 1;

    unsigned char *psc = lds;                   // [MS1][QPS]
    int16_t *mat = reinterpret_cast<int16_t *>(lds + ((MS1 * QPS + 7) & ~7));
    unsigned char *map = reinterpret_cast<unsigned char *>(mat + msize * msize);
    long long *ptab = reinterpret_cast<long long *>(map + 256 + ((8 - ((msize * msize * 2) & 7)) & 7));   // per pair: r offset, rlen, pair index
    // TR: eight steps of every lane's trace records are gathered in LDS and leave as one contiguous piece of the lane's stream
    // (96 or 128 bytes: whole 32-byte sectors; single 12/16-byte stores of thousands of resident lanes overflow the L2's write combining)
    constexpr int TSTG = PMX_QSTAGE;               // steps gathered per flush
    constexpr int TSTR = TSTG * TD + 1;              // dwords per lane (odd: the lanes' stores of a step fall into different banks)
    uint32_t *tstage = reinterpret_cast<uint32_t *>(ptab + 3 * NP) + (size_t)(wave * 64 + lane) * TSTR;

    const long long pair0 = (long long)blockIdx.x * NP;
    for (int i = tid; i < msize * msize; i += NT) mat[i] = gmat[i];
    for (int i = tid; i < 256; i += NT) map[i] = gmap[i];
    if (tid < NP) {
        long long pos = pair0 + tid; if (pos >= n) pos = n - 1;
        const long long pi = perm ? (long long)perm[pos] : pos;
        const long long rb = roff[pi];
        ptab[3 * tid + 0] = rb;
        ptab[3 * tid + 1] = roff[pi + 1] - rb;
        ptab[3 * tid + 2] = (pair0 + tid < n) ? pi : -1;
    }
    __syncthreads();
    // the shared byte profile: logical row er = l * R + k at byte l * RS + k; P virtual rows on top
    const int P = QP - qlen;
    // ROW OFFSET (round 3): every value of effective row er is stored + er * ext (on top of the column skew, which does the same
    // for E along a row), so the vertical gap needs no subtraction per row either: F(er) = max(F(er - 1), X(er - 1)) as it stands.
    // The diagonal step crosses one row: + ext in every profile byte; boundary values, hand-offs and captures carry the offset of
    // their row; the decisions compare values of one row and do not see it.  One instruction per row (of 7, or 15.5 with the trace).
    const int rx = ext;
    const int vrow_b = (row_pen ? 0 : open) + rx, vcol_b = (col_pen ? 0 : open) + rx;
    for (int er = tid; er < QP; er += NT) {
        unsigned char *sc = psc + (er / R) * RS + er % R;
        if (er >= P) {
            const int q = map[qbuf[er - P]];
            for (int sym = 0; sym < msize; ++sym) sc[sym * QPS] = (unsigned char)(mat[q * msize + sym] + open + rx);
            sc[msize * QPS] = (unsigned char)vcol_b;
        } else {
            for (int sym = 0; sym < msize; ++sym) sc[sym * QPS] = (unsigned char)vrow_b;
            sc[msize * QPS] = (unsigned char)(open + rx);   // virtual x virtual: score 0
        }
    }
    __syncthreads();

    const unsigned char *scL = psc + g * RS;
    const int rlA = (int)ptab[3 * pA + 1], rlB = (int)ptab[3 * pB + 1];
    const uint8_t *refA = rbuf + ptab[3 * pA + 0], *refB = rbuf + ptab[3 * pB + 0];
    auto fetch = [&](int x, int &ra, int &rb) {
        const int col = x - g;
        ra = (col >= 0 && col < rlA) ? (int)refA[col] : -1;
        rb = (col >= 0 && col < rlB) ? (int)refB[col] : -1;
    };
    auto sym_of = [&](int raw) -> int { return raw < 0 ? msize : (int)map[raw]; };

    auto pack2 = [](int a, int b) -> int { return (a & 0xFFFF) | (b << 16); };
    const int vExt = pack2(ext, ext), vC = pack2(open - ext, open - ext);
    const v2us one2 = {1, 1};
    const int base = nb + (G - g) * ext - open;
    auto left_h = [&](int erow) -> int { const int i = erow - P; return (i >= 0 && col_pen) ? -(open + i * ext) : 0; };
    auto below_f = [&](int erow) -> int { const int i = erow - P; return (i >= 0 && col_pen) ? -(open + i * ext) : -open; };

    int X[R], E[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { const int v = base + left_h(g * R + k) + (g * R + k) * rx; X[k] = pack2(v, v); E[k] = X[k]; }
    int Hout = X[R - 1];
    int Fout; { const int v = base + open + below_f((g + 1) * R) + ((g + 1) * R - 1) * rx; Fout = pack2(v, v); }     // (the running F of row er carries (er - 1) * ext)
    int diag0; { const int v = ((g == 0) ? base : base + left_h(g * R - 1)) + (g * R - 1) * rx; diag0 = pack2(v, v); }
    int topX = row_pen ? pack2(nb + (G + 1) * ext - 2 * open - rx, nb + (G + 1) * ext - 2 * open - rx)          // (row -1)
                       : pack2(nb + (G + 1) * ext - open - rx, nb + (G + 1) * ext - open - rx);
    const int topStep = row_pen ? 0 : vExt;
    int skewX = pack2((G - g + 1) * ext - open + (QP - 1) * rx, (G - g + 1) * ext - open + (QP - 1) * rx);    // (the last row's offset with it)
    int cb = 0;                                      // bias of the free-end captures (see pmx_nwsg16v_kernel), set once the wave's longest reference is known

    const v2s rl1 = PK(pack2(rlA - 1, rlB - 1)), rlv = PK(pack2(rlA, rlB));
    int jj = ((-g) & 0xFFFF) * 0x00010001;
    int res = 0;
    v2s bestrow = PK(0); int bestrowj = 0;
    v2s bestcol = PK(0); int bestcoli = 0;

    int w[2][2][RS / 4];
    auto load_scores = [&](int bsel, int symA, int symB) {
        const int *a = reinterpret_cast<const int *>(scL + symA * QPS), *b = reinterpret_cast<const int *>(scL + symB * QPS);
#pragma unroll
        for (int x = 0; x < RS / 4; ++x) { w[bsel][0][x] = a[x]; w[bsel][1][x] = b[x]; }
    };
    uint32_t *tw = TR ? tbuf + (((size_t)blockIdx.x * WAVES + wave) * 64 + lane) * (size_t)Tmax * TD : nullptr;
    auto step = [&](int bsel, int t) {
        const int Hin = n_shift_up<G>(Hout, topX, g);
        int F = n_shift_up<G>(Fout, topX, g);
        // rows in blocks of HBLK: the diagonal sums of a block are formed before its rows overwrite the strip (one block: halving it
        // for R = 20 did not lower the allocator's register count)
        constexpr int HBLK = R;
        int Tpre[HBLK];
        int tacc = 0, ty[TR ? (R + 1) / 2 : 1];
        int xcarry = diag0;                                  // previous column's H of the row above the block
#pragma unroll
        for (int k0 = 0; k0 < R; k0 += HBLK) {
        const int xnext = X[k0 + HBLK - 1];
#pragma unroll
        for (int k = k0; k < k0 + HBLK; ++k) {
            const int s = __builtin_amdgcn_perm(w[bsel][1][k / 4], w[bsel][0][k / 4], 0x0C000C00u | (unsigned)(k & 3) | ((4u + (unsigned)(k & 3)) << 16));
            Tpre[k - k0] = ((k == k0) ? xcarry : X[k - 1]) + s;
        }
        xcarry = xnext;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = k0; k < k0 + HBLK; ++k) {
            const int Fe = F;                                // (row offset: no subtraction)
            const int H = I32(n_max3f(PK(Tpre[k - k0]), PK(E[k]), PK(Fe)));
            const int Xn = H - vC;
            if (TR) {     // ND, NDL, EO, FO: the sign of each difference, inserted at its bit of the row pair's byte (see pmx_nwsg16v_kernel)
                const int dND = I32(PK(Tpre[k - k0]) - PK(H)), dNDL = I32(PK(Fe) - PK(H));
                const int dEO = I32(PK(E[k]) - PK(Xn)), dFO = I32(PK(Fe) - PK(Xn));
                if ((k & 1) == 0) {
                    tacc = dND;
                    tacc = s_bfi(0x40004000, dNDL, tacc);
                    tacc = s_bfi(0x20002000, dEO, tacc);
                    tacc = s_bfi(0x10001000, dFO, tacc);
                } else {
                    tacc = s_bfi(0x08000800, dND, tacc);
                    tacc = s_bfi(0x04000400, dNDL, tacc);
                    tacc = s_bfi(0x02000200, dEO, tacc);
                    tacc = s_bfi(0x01000100, dFO, tacc);
                    ty[k / 2] = tacc;                      // byte 1: pair A's rows k-1, k; byte 3: pair B's
                }
            }
            E[k] = I32(n_max3f(PK(E[k]), PK(Xn), PK(Xn)));
            F = I32(n_max3f(PK(Fe), PK(Xn), PK(Xn)));
            X[k] = Xn;
        }
        }
        if (TR) {
            if (R & 1) ty[R / 2] = tacc;                   // (odd R: the last row has a byte of its own; its low nibble is never read)
            // record: [A bytes 0 .. R/2-1][B bytes 0 .. R/2-1][pad]; u = [A_y A_y+1 B_y B_y+1] of two row pairs
            const int u01 = __builtin_amdgcn_perm(ty[0], ty[1], 0x03070105), u23 = __builtin_amdgcn_perm(ty[2], ty[3], 0x03070105);
            const int a0 = __builtin_amdgcn_perm(u01, u23, 0x01000504), b0 = __builtin_amdgcn_perm(u01, u23, 0x03020706);
            uint32_t *dst = tstage + (t & (TSTG - 1)) * TD;
            if (R >= 19) {
                // 10 + 10 bytes: a whole number of dwords (the 12-byte record of R = 10 carries two bytes of padding per 20 cells)
                const int u45 = __builtin_amdgcn_perm(ty[R >= 19 ? 4 : 0], ty[R >= 19 ? 5 : 0], 0x03070105), u67 = __builtin_amdgcn_perm(ty[R >= 19 ? 6 : 0], ty[R >= 19 ? 7 : 0], 0x03070105);
                const int u89 = __builtin_amdgcn_perm(ty[R >= 19 ? 8 : 0], ty[R >= 19 ? 9 : 0], 0x03070105);
                const int b1 = __builtin_amdgcn_perm(u45, u67, 0x03020706);
                dst[0] = (uint32_t)a0;                                                    // A0 A1 A2 A3
                dst[1] = (uint32_t)__builtin_amdgcn_perm(u45, u67, 0x01000504);           // A4 .. A7
                dst[2] = (uint32_t)__builtin_amdgcn_perm(u89, u01, 0x03020504);           // A8 A9 B0 B1
                dst[3] = (uint32_t)__builtin_amdgcn_perm(b0, b1, 0x01000706);             // B2 B3 B4 B5
                dst[TD - 1] = (uint32_t)__builtin_amdgcn_perm(b1, u89, 0x03020706);       // B6 B7 B8 B9
            } else if (R == 16) {
                const int u45 = __builtin_amdgcn_perm(ty[4], ty[R == 16 ? 5 : 0], 0x03070105), u67 = __builtin_amdgcn_perm(ty[R == 16 ? 6 : 0], ty[R == 16 ? 7 : 0], 0x03070105);
                dst[0] = (uint32_t)a0; dst[1] = (uint32_t)__builtin_amdgcn_perm(u45, u67, 0x01000504);
                dst[2] = (uint32_t)b0; dst[TD - 1] = (uint32_t)__builtin_amdgcn_perm(u45, u67, 0x03020706);
            } else {
                dst[0] = (uint32_t)a0;                                                    // A0 A1 A2 A3
                dst[1] = (uint32_t)__builtin_amdgcn_perm(ty[4], b0, 0x02010005);          // A4 B0 B1 B2
                dst[2] = (uint32_t)__builtin_amdgcn_perm(ty[4], b0, 0x0C0C0703);          // B3 B4 0 0
            }
        }
        diag0 = Hin;
        Hout = X[R - 1];
        Fout = F;
        // ---- captures (as in pmx_nwsg16v_kernel) ----
        const v2s jv = PK(jj);
        const int mLast = m_eq(jv, rl1);
        res = n_bfi(mLast, Hout, res);
        if (s2_end) {
            const v2s cand = PK(I32(__builtin_bit_cast(v2s, __builtin_bit_cast(v2us, Hout) - __builtin_bit_cast(v2us, skewX))));
            const int imp = m_lt(bestrow, cand) & m_ult(jv, rlv);
            bestrow = PK(n_bfi(imp, I32(cand), I32(bestrow)));
            bestrowj = n_bfi(imp, jj, bestrowj);
        }
        if (s1_end && __builtin_amdgcn_ballot_w64(mLast != 0) != 0) {
            v2s cm = PK(0); int krow = 0;
            v2s vals[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int er = g * R + k;
                vals[k] = PK(er >= P ? X[k] + pack2((QP - er) * rx, (QP - er) * rx) : 0);          // rows compare in the form of the last row's offset
                cm = n_max3f(cm, vals[k], vals[k]);
            }
#pragma unroll
            for (int k = R - 1; k >= 0; --k) {
                const int er = g * R + k;
                krow = n_bfi(m_eq(vals[k], cm), pack2(er, er), krow);
            }
            const int imp = m_lt(bestcol, cm) & mLast;
            bestcol = PK(n_bfi(imp, I32(cm), I32(bestcol)));
            bestcoli = n_bfi(imp, krow, bestcoli);
        }
        jj = I32(__builtin_bit_cast(v2s, __builtin_bit_cast(v2us, jj) + one2));
        skewX = I32(__builtin_bit_cast(v2s, __builtin_bit_cast(v2us, skewX) + __builtin_bit_cast(v2us, vExt)));
        topX += topStep;
    };

    int max_rlen = 0;
#pragma unroll
    for (int p = 0; p < NPW; ++p) max_rlen = max(max_rlen, (int)ptab[3 * (wave * NPW + p) + 1]);
    const int T = (max_rlen + G - 1 + 1) & ~1;
    cb = 4 * open + (QP + max_rlen + 2) * ext;
    skewX = pack2((G - g + 1) * ext - open + (QP - 1) * rx - cb, (G - g + 1) * ext - open + (QP - 1) * rx - cb);
    int r0a, r0b, r1a, r1b, m2a, m2b, m3a, m3b;
    fetch(0, r0a, r0b); fetch(1, r1a, r1b); fetch(2, m2a, m2b); fetch(3, m3a, m3b);
    load_scores(0, sym_of(r0a), sym_of(r0b));
    int nsA = sym_of(r1a), nsB = sym_of(r1b);
    for (int t = 0; t < T; t += 2) {
        load_scores(1, nsA, nsB);
        nsA = sym_of(m2a); nsB = sym_of(m2b);
        fetch(t + 4, m2a, m2b);
        __builtin_amdgcn_sched_barrier(0);
        step(0, t);
        __builtin_amdgcn_sched_barrier(0);
        load_scores(0, nsA, nsB);
        nsA = sym_of(m3a); nsB = sym_of(m3b);
        fetch(t + 5, m3a, m3b);
        __builtin_amdgcn_sched_barrier(0);
        step(1, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (TR && ((t & (TSTG - 2)) == (TSTG - 2) || t + 2 >= T)) {          // TSTG steps gathered (or the sweep ends)
            // The wave's 64 pieces of lane stream (TSTG * TD dwords each, contiguous in HBM) leave together: item w = 16-byte piece
            // w % PPS of lane stream w / PPS, so neighbouring lanes write neighbouring bytes and every store covers whole stretches
            constexpr int PPS = TSTG * TD / 4;                            // 16-byte pieces per stream and flush
            const uint32_t *wstage = tstage - (size_t)lane * TSTR;        // the wave's staging area
            uint32_t *wtw = tw - (size_t)lane * Tmax * TD;                // the wave's lane stream 0
#pragma unroll
            for (int x = 0; x < PPS; ++x) {
                const int w = x * 64 + lane, stream = w / PPS, piece = w - stream * PPS;
                const uint32_t *src = wstage + stream * TSTR + piece * 4;
                uint4 v; v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
                *reinterpret_cast<uint4 *>(wtw + ((size_t)stream * Tmax + (size_t)(t & ~(TSTG - 1))) * TD + piece * 4) = v;
            }
        }
    }

    unsigned keyA = ((unsigned)(I32(bestcol) & 0xFFFF) << 16) | (0xFFFFu - (unsigned)(bestcoli & 0xFFFF));
    unsigned keyB = ((unsigned)((unsigned)I32(bestcol) >> 16) << 16) | (0xFFFFu - ((unsigned)bestcoli >> 16));
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) {
        const unsigned oa = __shfl_xor(keyA, off, 64), ob = __shfl_xor(keyB, off, 64);
        keyA = oa > keyA ? oa : keyA;
        keyB = ob > keyB ? ob : keyB;
    }
    const int lastlane = slotw * G + G - 1;
    const int resL = __shfl(res, lastlane, 64);
    const int browL = __shfl(I32(bestrow), lastlane, 64), browjL = __shfl(bestrowj, lastlane, 64);
    if (g == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long pi = ptab[3 * (h ? pB : pA) + 2];
            if (pi >= 0) {
                const int ql = qlen, rl = h ? rlB : rlA;
                const int unsk = nb + (rl - 1 + G) * ext - open + ext;
                const int corner = (int)(h ? ((unsigned)resL >> 16) : (resL & 0xFFFF)) - unsk - (QP - 1) * rx;
                pmx_record_t rec;
                rec.flags = 0;
                if (!s1_end && !s2_end) { rec.score = corner; rec.end_query = ql - 1; rec.end_ref = rl - 1; }
                else {
                    int best = -2147483647 - 1, ei = 0, ej = 0;
                    if (s2_end) {
                        best = (int)(h ? ((unsigned)browL >> 16) : (browL & 0xFFFF)) - nb - cb;
                        ei = ql - 1; ej = (int)(h ? ((unsigned)browjL >> 16) : (browjL & 0xFFFF));
                    }
                    if (s1_end) {
                        const unsigned key = h ? keyB : keyA;
                        const int cv = (int)(key >> 16) - unsk - QP * rx;
                        if (cv > best) { best = cv; ei = (int)(0xFFFFu - (key & 0xFFFFu)) - P; ej = rl - 1; }
                    }
                    rec.score = best; rec.end_query = ei; rec.end_ref = ej;
                }
                out[pi] = rec;
            }
        }
    }
}

// Per-pair queries over a large alphabet (protein all-vs-all style batches): no LDS profile, the scores are byte reads
// from the transposed matrix in LDS (see pmx_sw16m.hip); column msize of that matrix is the virtual row, row msize
// the virtual / pad column, so the boundary scores of the second generation come out of the same lookup.
template <int G, int R, bool TR>
__global__ __launch_bounds__(64)
void pmx_nwsg16m_kernel(const uint8_t *__restrict__ qbuf, const int64_t *__restrict__ qoff,
                        const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff,
                        long long n, const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap,
                        int msize, int open, int ext,
                        int col_pen, int row_pen, int s1_end, int s2_end, int nb,
                        const unsigned *__restrict__ perm,
                        pmx_record_t *__restrict__ out, uint32_t *__restrict__ tbuf, int Tmax)
{
    static_assert(!TR || R == 16, "trace: four packed planes of 4 rows");
    constexpr int QP = G * R;
    constexpr int NPW = 2 * (64 / G);
    constexpr int MSTR = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int lane = threadIdx.x;
    const int g = lane % G, slotw = lane / G;
    const int pA = 2 * slotw, pB = pA + 1;

    // matT[r][q] (bytes): letters: score + open; column msize = a virtual row; row msize = the pad symbol (virtual column)
    unsigned char *matT = lds;
    unsigned char *map = lds + (msize + 1) * MSTR;
    long long *ptab = reinterpret_cast<long long *>(map + 256);       // per pair: q offset, qlen, r offset, rlen, pair index
    const int rx = ext;                               // row offset (see ROWX at pmx_nwsg16v_kernel): every score byte + ext, no F - ext per row
    const int vrow_b = (row_pen ? 0 : open) + rx, vcol_b = (col_pen ? 0 : open) + rx;
    const long long pair0 = (long long)blockIdx.x * NPW;
    for (int i = lane; i < (msize + 1) * MSTR; i += 64) {
        const int r = i / MSTR, q = i % MSTR;
        int v;
        if (r < msize) v = q < msize ? gmat[q * msize + r] + open + rx : vrow_b;
        else v = q < msize ? vcol_b : open + rx;                      // virtual x virtual: score 0
        matT[i] = (unsigned char)v;
    }
    for (int i = lane; i < 256; i += 64) map[i] = gmap[i];
    if (lane < NPW) {
        long long pos = pair0 + lane; if (pos >= n) pos = n - 1;
        const long long pi = perm ? (long long)perm[pos] : pos;
        const long long qb = qoff[pi], rb = roff[pi];
        ptab[5 * lane + 0] = qb; ptab[5 * lane + 1] = qoff[pi + 1] - qb;
        ptab[5 * lane + 2] = rb; ptab[5 * lane + 3] = roff[pi + 1] - rb;
        ptab[5 * lane + 4] = (pair0 + lane < n) ? pi : -1;
    }
    __syncthreads();

    const int qlA = (int)ptab[5 * pA + 1], qlB = (int)ptab[5 * pB + 1];
    const int PvA = QP - qlA, PvB = QP - qlB;                         // virtual rows on top (query bottom-aligned)
    const int rlA = (int)ptab[5 * pA + 3], rlB = (int)ptab[5 * pB + 3];
    const uint8_t *refA = rbuf + ptab[5 * pA + 2], *refB = rbuf + ptab[5 * pB + 2];
    int qa[R], qb_[R];                                                // LDS offsets of this lane's rows inside a matT row
    {
        const uint8_t *qA = qbuf + ptab[5 * pA + 0], *qB = qbuf + ptab[5 * pB + 0];
        unsigned char ra[R], rb[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int er = g * R + k;
            ra[k] = er >= PvA ? qA[er - PvA] : (unsigned char)0;
            rb[k] = er >= PvB ? qB[er - PvB] : (unsigned char)0;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int er = g * R + k;
            qa[k] = er >= PvA ? (int)map[ra[k]] : msize;
            qb_[k] = er >= PvB ? (int)map[rb[k]] : msize;
        }
    }
    auto fetch = [&](int x, int &ra, int &rb) {
        const int col = x - g;
        ra = (col >= 0 && col < rlA) ? (int)refA[col] : -1;
        rb = (col >= 0 && col < rlB) ? (int)refB[col] : -1;
    };
    auto sym_of = [&](int raw) -> int { return (raw < 0 ? msize : (int)map[raw]) * MSTR; };   // byte offset of the matT row

    auto pack2 = [](int a, int b) -> int { return (a & 0xFFFF) | (b << 16); };
    const int vExt = pack2(ext, ext), vC = pack2(open - ext, open - ext);
    const v2us one2 = {1, 1};
    const int base = nb + (G - g) * ext - open;
    auto left_h = [&](int erow, int P) -> int { const int i = erow - P; return (i >= 0 && col_pen) ? -(open + i * ext) : 0; };
    auto below_f = [&](int erow, int P) -> int { const int i = erow - P; return (i >= 0 && col_pen) ? -(open + i * ext) : -open; };

    int X[R], E[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { const int ro = (g * R + k) * rx; X[k] = pack2(base + left_h(g * R + k, PvA) + ro, base + left_h(g * R + k, PvB) + ro); E[k] = X[k]; }
    const int roF = ((g + 1) * R - 1) * rx, roD = (g * R - 1) * rx, roL = (QP - 1) * rx;
    int Hout = X[R - 1];
    int Fout = pack2(base + open + below_f((g + 1) * R, PvA) + roF, base + open + below_f((g + 1) * R, PvB) + roF);
    int diag0 = (g == 0) ? pack2(base + roD, base + roD) : pack2(base + left_h(g * R - 1, PvA) + roD, base + left_h(g * R - 1, PvB) + roD);
    int topX = row_pen ? pack2(nb + (G + 1) * ext - 2 * open - rx, nb + (G + 1) * ext - 2 * open - rx)
                       : pack2(nb + (G + 1) * ext - open - rx, nb + (G + 1) * ext - open - rx);
    const int topStep = row_pen ? 0 : vExt;
    int skewX = pack2((G - g + 1) * ext - open + roL, (G - g + 1) * ext - open + roL);
    int cb = 0;                                      // bias of the free-end captures (see pmx_nwsg16v_kernel)

    const v2s rl1 = PK(pack2(rlA - 1, rlB - 1)), rlv = PK(pack2(rlA, rlB));
    int jj = ((-g) & 0xFFFF) * 0x00010001;
    int res = 0;
    v2s bestrow = PK(0); int bestrowj = 0;
    v2s bestcol = PK(0); int bestcoli = 0;

    int w[2][R];
    auto load_scores = [&](int bsel, int rowA, int rowB) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int sa = matT[rowA + qa[k]], sb = matT[rowB + qb_[k]];
            w[bsel][k] = sa | (sb << 16);
        }
    };
    // trace records of 16 bytes per lane and step, lane-major: every lane's steps are contiguous, so what the walk reads along a row
    // or a diagonal sits in one cache line (consecutive stores of a lane fill its 128-byte lines in L2)
    const size_t t_ss = 4;
    uint32_t *tw = TR ? tbuf + ((size_t)blockIdx.x * Tmax) * 256 + (size_t)lane * Tmax * 4 : nullptr;
    auto push = [&](int &pl, int a, int b) {          // pl = 2 * pl + (a < b), per half
        const v2us fifteen = {15, 15};
        const int bit = I32(__builtin_bit_cast(v2s, __builtin_bit_cast(v2us, PK(a) - PK(b)) >> fifteen));
        int r;
        asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(pl), "v"(0x00020002), "v"(bit));
        pl = r;
    };
    auto step = [&](int bsel, int t) {
        const int Hin = n_shift_up<G>(Hout, topX, g);
        int F = n_shift_up<G>(Fout, topX, g);
        int Tpre[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            Tpre[k] = ((k == 0) ? diag0 : X[k - 1]) + w[bsel][k];
        }
        __builtin_amdgcn_sched_barrier(0);
        int plane[TR ? R / 4 : 1];
        if (TR) {
#pragma unroll
            for (int x = 0; x < R / 4; ++x) plane[x] = 0;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int Fe = F;                                // (row offset: no subtraction)
            const int H = I32(n_max3f(PK(Tpre[k]), PK(E[k]), PK(Fe)));
            const int Xn = H - vC;
            if (TR) {
                push(plane[k / 4], Tpre[k], H);      // ND
                push(plane[k / 4], Fe, H);           // NDL
                push(plane[k / 4], E[k], Xn);        // EO
                push(plane[k / 4], Fe, Xn);          // FO
            }
            E[k] = I32(n_max3f(PK(E[k]), PK(Xn), PK(Xn)));
            F = I32(n_max3f(PK(Fe), PK(Xn), PK(Xn)));
            X[k] = Xn;
        }
        if (TR) {
            uint4 w4;
            w4.x = __builtin_amdgcn_perm(plane[0], plane[1], 0x00010405);
            w4.y = __builtin_amdgcn_perm(plane[2], plane[3], 0x00010405);
            w4.z = __builtin_amdgcn_perm(plane[0], plane[1], 0x02030607);
            w4.w = __builtin_amdgcn_perm(plane[2], plane[3], 0x02030607);
            *reinterpret_cast<uint4 *>(tw + (size_t)t * t_ss) = w4;
        }
        diag0 = Hin;
        Hout = X[R - 1];
        Fout = F;
        // ---- captures (as in pmx_nwsg16v_kernel) ----
        const v2s jv = PK(jj);
        const int mLast = m_eq(jv, rl1);
        res = n_bfi(mLast, Hout, res);
        if (s2_end) {
            const v2s cand = PK(I32(__builtin_bit_cast(v2s, __builtin_bit_cast(v2us, Hout) - __builtin_bit_cast(v2us, skewX))));
            const int imp = m_lt(bestrow, cand) & m_ult(jv, rlv);
            bestrow = PK(n_bfi(imp, I32(cand), I32(bestrow)));
            bestrowj = n_bfi(imp, jj, bestrowj);
        }
        if (s1_end && __builtin_amdgcn_ballot_w64(mLast != 0) != 0) {
            v2s cm = PK(0); int krow = 0;
            v2s vals[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int er = g * R + k;
                const int mreal = ~m_lt(PK(pack2(er, er)), PK(pack2(PvA, PvB)));
                vals[k] = PK((X[k] + pack2((QP - er) * rx, (QP - er) * rx)) & mreal);
                cm = n_max3f(cm, vals[k], vals[k]);
            }
#pragma unroll
            for (int k = R - 1; k >= 0; --k) {
                const int er = g * R + k;
                krow = n_bfi(m_eq(vals[k], cm), pack2(er, er), krow);
            }
            const int imp = m_lt(bestcol, cm) & mLast;
            bestcol = PK(n_bfi(imp, I32(cm), I32(bestcol)));
            bestcoli = n_bfi(imp, krow, bestcoli);
        }
        jj = I32(__builtin_bit_cast(v2s, __builtin_bit_cast(v2us, jj) + one2));
        skewX = I32(__builtin_bit_cast(v2s, __builtin_bit_cast(v2us, skewX) + __builtin_bit_cast(v2us, vExt)));
        topX += topStep;
    };

    int max_rlen = 0;
#pragma unroll
    for (int p = 0; p < NPW; ++p) max_rlen = max(max_rlen, (int)ptab[5 * p + 3]);
    const int T = (max_rlen + G - 1 + 1) & ~1;
    cb = 4 * open + (QP + max_rlen + 2) * ext;
    skewX = pack2((G - g + 1) * ext - open + roL - cb, (G - g + 1) * ext - open + roL - cb);
    int r0a, r0b, r1a, r1b, m2a, m2b, m3a, m3b;
    fetch(0, r0a, r0b); fetch(1, r1a, r1b); fetch(2, m2a, m2b); fetch(3, m3a, m3b);
    load_scores(0, sym_of(r0a), sym_of(r0b));
    int nsA = sym_of(r1a), nsB = sym_of(r1b);
    for (int t = 0; t < T; t += 2) {
        load_scores(1, nsA, nsB);
        nsA = sym_of(m2a); nsB = sym_of(m2b);
        fetch(t + 4, m2a, m2b);
        __builtin_amdgcn_sched_barrier(0);
        step(0, t);
        __builtin_amdgcn_sched_barrier(0);
        load_scores(0, nsA, nsB);
        nsA = sym_of(m3a); nsB = sym_of(m3b);
        fetch(t + 5, m3a, m3b);
        __builtin_amdgcn_sched_barrier(0);
        step(1, t + 1);
        __builtin_amdgcn_sched_barrier(0);
    }

    unsigned keyA = ((unsigned)(I32(bestcol) & 0xFFFF) << 16) | (0xFFFFu - (unsigned)(bestcoli & 0xFFFF));
    unsigned keyB = ((unsigned)((unsigned)I32(bestcol) >> 16) << 16) | (0xFFFFu - ((unsigned)bestcoli >> 16));
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) {
        const unsigned oa = __shfl_xor(keyA, off, 64), ob = __shfl_xor(keyB, off, 64);
        keyA = oa > keyA ? oa : keyA;
        keyB = ob > keyB ? ob : keyB;
    }
    const int lastlane = slotw * G + G - 1;
    const int resL = __shfl(res, lastlane, 64);
    const int browL = __shfl(I32(bestrow), lastlane, 64), browjL = __shfl(bestrowj, lastlane, 64);
    if (g == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long pi = ptab[5 * (h ? pB : pA) + 4];
            if (pi >= 0) {
                const int ql = h ? qlB : qlA, rl = h ? rlB : rlA, P = h ? PvB : PvA;
                const int unsk = nb + (rl - 1 + G) * ext - open + ext;
                const int corner = (int)(h ? ((unsigned)resL >> 16) : (resL & 0xFFFF)) - unsk - roL;
                pmx_record_t rec;
                rec.flags = 0;
                if (!s1_end && !s2_end) { rec.score = corner; rec.end_query = ql - 1; rec.end_ref = rl - 1; }
                else {
                    int best = -2147483647 - 1, ei = 0, ej = 0;
                    if (s2_end) {
                        best = (int)(h ? ((unsigned)browL >> 16) : (browL & 0xFFFF)) - nb - cb;
                        ei = ql - 1; ej = (int)(h ? ((unsigned)browjL >> 16) : (browjL & 0xFFFF));
                    }
                    if (s1_end) {
                        const unsigned key = h ? keyB : keyA;
                        const int cv = (int)(key >> 16) - unsk - QP * rx;
                        if (cv > best) { best = cv; ei = (int)(0xFFFFu - (key & 0xFFFFu)) - P; ej = rl - 1; }
                    }
                    rec.score = best; rec.end_query = ei; rec.end_ref = ej;
                }
                out[pi] = rec;
            }
        }
    }
}

template <int G, int R, bool TR = false>
static int launch_nwsgm(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext, int nb,
                        pmx_record_t *d_out, hipStream_t stream, uint32_t *tbuf = nullptr, int Tmax = 0)
{
    constexpr int NP = 2 * (64 / G);
    const size_t lds = (size_t)(m.msize + 1) * 32 + 256 + (size_t)NP * 40;
    const bool sg = mode == PMX_MODE_SG;
    const int col_pen = !(sg && (sg_flags & PMX_SG_QB)), row_pen = !(sg && (sg_flags & PMX_SG_DB));
    const int s1_end = sg && (sg_flags & PMX_SG_QE), s2_end = sg && (sg_flags & PMX_SG_DE);
    const long long blocks = (b.n + NP - 1) / NP;
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL((pmx_nwsg16m_kernel<G, R, TR>), dim3((unsigned)blocks), dim3(64), lds, stream,
                       b.qbuf, b.qoff, b.rbuf, b.roff, (long long)b.n, m.scores, m.mapper, m.msize, open, ext,
                       col_pen, row_pen, s1_end ? 1 : 0, s2_end ? 1 : 0, nb, b.perm, d_out, tbuf, Tmax);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

template <int G, int R, bool TR = false>
static int launch_nwsgq(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext, int nb,
                        pmx_record_t *d_out, hipStream_t stream, uint32_t *tbuf = nullptr, int Tmax = 0)
{
    constexpr int RS = (R + 3) / 4 * 4, WAVES = 4, NP = 2 * (64 / G) * WAVES;
    const size_t lds = (size_t)(m.msize + 1) * G * RS + 8 + (size_t)m.msize * m.msize * 2 + 256 + 8 + (size_t)NP * 24 +
                       (TR ? (size_t)WAVES * 64 * (PMX_QSTAGE * ((R + 3) / 4) + 1) * 4 : 0);
    if (lds > 160 * 1024) return 1;
    const bool sg = mode == PMX_MODE_SG;
    const int col_pen = !(sg && (sg_flags & PMX_SG_QB)), row_pen = !(sg && (sg_flags & PMX_SG_DB));
    const int s1_end = sg && (sg_flags & PMX_SG_QE), s2_end = sg && (sg_flags & PMX_SG_DE);
    const long long blocks = (b.n + NP - 1) / NP;
    if (blocks <= 0) return 0;
    if constexpr (TR && R >= 19) {
        if (!s1_end && !s2_end && !pmx_env("PMX_NWSGQ_ENDS_ALWAYS")) {       // no free end: the instance without captures (three waves per SIMD)
            { const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_nwsg16q_kernel<G, R, WAVES, TR, false>)); if (rc) return rc; }
            hipLaunchKernelGGL((pmx_nwsg16q_kernel<G, R, WAVES, TR, false>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, stream,
                               b.qbuf, b.q_shared, b.rbuf, b.roff, (long long)b.n, m.scores, m.mapper, m.msize, open, ext,
                               col_pen, row_pen, 0, 0, nb, b.perm, d_out, tbuf, Tmax);
            const hipError_t e = hipGetLastError();
            return e == hipSuccess ? 0 : -(int)e;
        }
    }
    { const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_nwsg16q_kernel<G, R, WAVES, TR>)); if (rc) return rc; }
    hipLaunchKernelGGL((pmx_nwsg16q_kernel<G, R, WAVES, TR>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, stream,
                       b.qbuf, b.q_shared, b.rbuf, b.roff, (long long)b.n, m.scores, m.mapper, m.msize, open, ext,
                       col_pen, row_pen, s1_end ? 1 : 0, s2_end ? 1 : 0, nb, b.perm, d_out, tbuf, Tmax);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// Traceback with a shared query (profile arm): shapes <16,10> <16,16> <32,10> <32,16> <64,16>, one virtual row at least on top
// (the walk needs row -1), bounded differences for the one-instruction decision merge.  *variant = 30 + shape index.
// <16,20>: 320 rows like <32,10>, but twice the rows per lane -- the per-step work that is not per row (hand-off, captures, record
// assembly, staging flush) is shared by twice the cells -- and a record of exactly 20 bytes instead of 12 for half the cells.
// (<16,19>, round 4: a 300-row query -- the typical protein -- fills 300 of 304 rows instead of 300 of 320; tried before <16,20>)
static const int kQShapeG[7] = {16, 16, 16, 32, 32, 64, 16}, kQShapeR[7] = {10, 16, 20, 10, 16, 16, 19}, kQShapeOrder[7] = {0, 1, 6, 2, 3, 4, 5};
int pmx_nwsgq_trace_plan(const PmxBatch &b, const PmxDevMatrix &m, int mode, int open, int ext,
                         int *variant, int *Tmax, size_t *trace_bytes, int *G_out, int *R_out, int short_waves)
{
    if (mode != PMX_MODE_NW && mode != PMX_MODE_SG) return 1;
    if (!b.q_shared || !pmx_nwsgv_bias(b, m, open, ext, 1)) return 1;
    if ((m.max > 0 ? m.max : 0) + 2 * open > 250) return 1;
    for (int vo = 0; vo < 7; ++vo) {
        const int v = kQShapeOrder[vo];
        const int G = kQShapeG[v], R = kQShapeR[v];
        if (b.q_shared > G * R - 1) continue;
        if (R >= 19 && (short_waves || pmx_env("PMX_NWSGQ_NO_R20"))) continue;      // (short_waves: half the rows per lane = half the time per wave)
        if (R == 19 && pmx_env("PMX_NWSGQ_NO_R19")) continue;
        const size_t lds = (size_t)(m.msize + 1) * G * ((R + 3) / 4 * 4) + 8 + (size_t)m.msize * m.msize * 2 + 256 + 8 + (size_t)(2 * (64 / G) * 4) * 24 +
                           (size_t)4 * 64 * (PMX_QSTAGE * ((R + 3) / 4) + 1) * 4;
        if (lds > 160 * 1024) continue;
        const long long NP = 2 * (64 / G) * 4;
        *variant = 30 + v; *G_out = G; *R_out = R;
        *Tmax = (b.max_rlen + G - 1 + 1 + 15) & ~15;
        *trace_bytes = (size_t)((b.n + NP - 1) / NP) * 4 * (size_t)*Tmax * 64 * (size_t)((R + 3) / 4) * 4;
        return 0;
    }
    return 1;
}

// Pairs that are resident at once (workgroups per CU x CUs x pairs per workgroup) in the sweep `variant` would launch: waves of one
// launch over equally long references all take the same time, so a launch of N workgroups runs for ceil(N / resident) "rounds" --
// the caller sizes its chunks in whole rounds (pmx_api.hip, stats_by_trace_shared).  0: unknown.
template <int G, int R, bool ENDS>
static long long nwsgq_round_pairs(size_t lds)
{
    static int cus = 0;
    if (!cus) { int dev = 0; hipDeviceProp_t p; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0; cus = p.multiProcessorCount; }
    if (pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_nwsg16q_kernel<G, R, 4, true, ENDS>))) return 0;
    int nblk = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, reinterpret_cast<const void *>(&pmx_nwsg16q_kernel<G, R, 4, true, ENDS>), 256, lds) != hipSuccess || nblk <= 0) return 0;
    return (long long)nblk * cus * (2 * (64 / G) * 4);
}
long long pmx_nwsgq_trace_round_pairs(int variant, const PmxDevMatrix &m, int mode, int sg_flags)
{
    const int v = variant - 30;
    if (v < 0 || v >= 7) return 0;
    const int G = kQShapeG[v], R = kQShapeR[v];
    const size_t lds = (size_t)(m.msize + 1) * G * ((R + 3) / 4 * 4) + 8 + (size_t)m.msize * m.msize * 2 + 256 + 8 + (size_t)(2 * (64 / G) * 4) * 24 +
                       (size_t)4 * 64 * (PMX_QSTAGE * ((R + 3) / 4) + 1) * 4;
    const bool ends = mode == PMX_MODE_SG && (sg_flags & (PMX_SG_QE | PMX_SG_DE));
    switch (v) {
    case 0: return nwsgq_round_pairs<16, 10, true>(lds);
    case 1: return nwsgq_round_pairs<16, 16, true>(lds);
    case 2: return (ends || pmx_env("PMX_NWSGQ_ENDS_ALWAYS")) ? nwsgq_round_pairs<16, 20, true>(lds) : nwsgq_round_pairs<16, 20, false>(lds);
    case 3: return nwsgq_round_pairs<32, 10, true>(lds);
    case 4: return nwsgq_round_pairs<32, 16, true>(lds);
    case 5: return nwsgq_round_pairs<64, 16, true>(lds);
    case 6: return (ends || pmx_env("PMX_NWSGQ_ENDS_ALWAYS")) ? nwsgq_round_pairs<16, 19, true>(lds) : nwsgq_round_pairs<16, 19, false>(lds);
    }
    return 0;
}

int pmx_launch_nwsgq_trace(int variant, const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                           pmx_record_t *d_out, uint32_t *tbuf, int Tmax, hipStream_t stream)
{
    const int nb = pmx_nwsgv_bias(b, m, open, ext, 1);
    if (!nb) return 1;
    switch (variant - 30) {
    case 0: return launch_nwsgq<16, 10, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    case 1: return launch_nwsgq<16, 16, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    case 2: return launch_nwsgq<16, 20, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    case 3: return launch_nwsgq<32, 10, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    case 4: return launch_nwsgq<32, 16, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    case 5: return launch_nwsgq<64, 16, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    case 6: return launch_nwsgq<16, 19, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    }
    return 1;
}

// ------------------------------------------------------------------------ host side ----
template <int G, int R>
static int launch_nwsg(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                       pmx_record_t *d_out, hipStream_t stream)
{
    constexpr int QP = G * R, NP = 2 * (64 / G);
    const int RP = ((b.max_rlen + 2 * (G - 1) + 4 + 7) / 4) * 4;
    const size_t lds = (size_t)NP * (m.msize + 1) * QP * 2 + (size_t)NP * RP +
                       (size_t)m.msize * m.msize * 2 + 256 + 8 + (size_t)NP * 40;
    if (lds > 160 * 1024) return 1;
    { const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_nwsg16_kernel<G, R>)); if (rc) return rc; }
    const bool sg = mode == PMX_MODE_SG;
    const int col_pen = !(sg && (sg_flags & PMX_SG_QB)), row_pen = !(sg && (sg_flags & PMX_SG_DB));
    const int s1_end = sg && (sg_flags & PMX_SG_QE), s2_end = sg && (sg_flags & PMX_SG_DE);
    const long long blocks = (b.n + NP - 1) / NP;
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL((pmx_nwsg16_kernel<G, R>), dim3((unsigned)blocks), dim3(64), lds, stream,
                       b.qbuf, b.qoff, b.rbuf, b.roff, (long long)b.n, m.scores, m.mapper,
                       m.msize, open, ext, RP, b.q_shared, col_pen, row_pen, s1_end ? 1 : 0, s2_end ? 1 : 0, b.perm, d_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

static thread_local bool g_nwsgv_pt = false;          // the last launch_nwsgv of this thread ran the perm-table form first (kernel names)
// PTOK: the shape has a perm-table instantiation (the ones BASELINE-sized DNA batches take; every instantiation costs compile time)
template <int G, int R, bool TR> struct NwsgPtShape {
    static constexpr bool value = (!TR && G == 8 && (R == 7 || R == 10 || R == 13 || R == 16 || R == 19 || R == 20)) || (G == 16 && R == 16) ||
                                  (!TR && G == 32 && R == 16);
};

template <int G, int R, bool TR = false, bool FETCH = false, bool TRB = false>
static int launch_nwsgv(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext, int nb,
                        pmx_record_t *d_out, hipStream_t stream, uint32_t *tbuf = nullptr, int Tmax = 0)
{
    // the window proof once more with THIS shape's rows (round-3 advice: the estimate inside pmx_nwsgv_bias can be smaller than G * R)
    if (!b.track8 && !pmx_nwsgv_bias(b, m, open, ext, 1, G * R)) return 1;
    if (TR && !TRB && (m.max > 0 ? m.max : 0) + 2 * open <= 250 && !pmx_env("PMX_TRACE_NO_BFI"))     // bounded differences: the one-instruction merge
        return launch_nwsgv<G, R, TR, FETCH, TR>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    constexpr int RS = (R + 3) / 4 * 4, NP = 2 * (64 / G);
    const int RP = ((b.max_rlen + 2 * (G - 1) + 4 + 7) / 4) * 4;
    const size_t lds = (size_t)NP * (m.msize + 1) * G * RS + (FETCH ? 0 : (size_t)NP * RP) +
                       (size_t)m.msize * m.msize * 2 + 256 + 8 + (size_t)NP * 40;
    if (lds > 160 * 1024) return 1;
    { const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_nwsg16v_kernel<G, R, TR, FETCH, TRB>)); if (rc) return rc; }
    const bool sg = mode == PMX_MODE_SG;
    const int col_pen = !(sg && (sg_flags & PMX_SG_QB)), row_pen = !(sg && (sg_flags & PMX_SG_DB));
    const int s1_end = sg && (sg_flags & PMX_SG_QE), s2_end = sg && (sg_flags & PMX_SG_DE);
    const long long blocks = (b.n + NP - 1) / NP;
    if (blocks <= 0) return 0;
    if constexpr (!TR) {
        if (b.track8) {
            // width 8: the range of H decides the saturation flag.  Blocks whose pairs all saturate by a penalised boundary alone (most
            // reads: -(open + 62 extend) < -128 under 5 / 2) need no tracking: they run on the perm-table form with its row offset, at
            // width 16's speed; that launch marks the others, and the tracking form (no row offset: it compares H across rows in
            // every step) runs over exactly those.
            const int *only8 = nullptr;
            g_nwsgv_pt = false;
            if constexpr (NwsgPtShape<G, R, TR>::value && !FETCH) {
                const int nb_rowx = pmx_nwsgv_bias(b, m, open, ext, 1);      // (the caller's nb is the tracking form's: no row offset)
                if (nb_rowx && m.msize <= 5 && b.blockflag && !b.q_shared && (col_pen || row_pen) && !pmx_env("PMX_NWSG16_NO_PERMTABLE")) {
                    const size_t lds_pt = (size_t)NP * RP + (size_t)m.msize * m.msize * 2 + 256 + 8 + (size_t)NP * 40 + 160 + (size_t)NP * ((G * R + 3) / 4 * 4) + 32;
                  if (lds_pt <= 160 * 1024 && (lds_pt <= 48 * 1024 || pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_nwsg16v_kernel<G, R, TR, false, TRB, true>)) == 0)) {
                    hipLaunchKernelGGL((pmx_nwsg16v_kernel<G, R, TR, false, TRB, true>), dim3((unsigned)blocks), dim3(64), lds_pt, stream,
                                       b.qbuf, b.qoff, b.rbuf, b.roff, (long long)b.n, m.scores, m.mapper,
                                       m.msize, open, ext, RP, b.q_shared, col_pen, row_pen, s1_end ? 1 : 0, s2_end ? 1 : 0, nb_rowx, b.perm, d_out, tbuf, Tmax,
                                       1, b.blockflag, (const int *)nullptr);
                    const hipError_t e = hipGetLastError();
                    if (e != hipSuccess) return -(int)e;
                    only8 = b.blockflag;
                    g_nwsgv_pt = true;
                  }       // (else: the staged references do not fit beside the tables -- the LDS-profile form takes every block)
                }
            }
            { const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_nwsg16v_kernel<G, R, TR, FETCH, TRB, false, false>)); if (rc) return rc; }
            if (!only8 && b.blockflag) { const hipError_t e = hipMemsetAsync(b.blockflag, 0xFF, (size_t)blocks * sizeof(int), stream); if (e != hipSuccess) return -(int)e; }
            hipLaunchKernelGGL((pmx_nwsg16v_kernel<G, R, TR, FETCH, TRB, false, false>), dim3((unsigned)blocks), dim3(64), lds, stream,
                               b.qbuf, b.qoff, b.rbuf, b.roff, (long long)b.n, m.scores, m.mapper,
                               m.msize, open, ext, RP, b.q_shared, col_pen, row_pen, s1_end ? 1 : 0, s2_end ? 1 : 0, nb, b.perm, d_out, tbuf, Tmax,
                               b.track8, (int *)nullptr, only8);
            const hipError_t e = hipGetLastError();
            return e == hipSuccess ? 0 : -(int)e;
        }
    }
    // Alphabets of <= 4 letters (+ wildcard): the perm-table form first (no LDS profile: see PT at the kernel); it marks the blocks
    // it cannot take -- query lengths that differ inside the block, a query letter beyond the first four -- and the LDS-profile form
    // below then runs over exactly those.  Needs the caller's per-block flags (PmxBatch::blockflag).
    const int *only = nullptr;
    g_nwsgv_pt = false;
    if constexpr (NwsgPtShape<G, R, TR>::value && !FETCH && (!TR || TRB)) {
        if (m.msize <= 5 && b.blockflag && !b.track8 && !b.q_shared && !pmx_env("PMX_NWSG16_NO_PERMTABLE")) {
            const size_t lds_pt = (size_t)NP * RP + (size_t)m.msize * m.msize * 2 + 256 + 8 + (size_t)NP * 40 + 160 + (size_t)NP * ((G * R + 3) / 4 * 4) + 32 +
                                  (TR ? (size_t)64 * 33 * 4 : 0);
            // (more than the default 48 KB of dynamic LDS -- short queries against references of several kbp -- needs the opt-in; where the
            //  staged references do not fit at all, or the opt-in fails, the LDS-profile form below takes every block)
            if (lds_pt <= 160 * 1024 && (lds_pt <= 48 * 1024 || pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_nwsg16v_kernel<G, R, TR, false, TRB, true>)) == 0)) {
                hipLaunchKernelGGL((pmx_nwsg16v_kernel<G, R, TR, false, TRB, true>), dim3((unsigned)blocks), dim3(64), lds_pt, stream,
                                   b.qbuf, b.qoff, b.rbuf, b.roff, (long long)b.n, m.scores, m.mapper,
                                   m.msize, open, ext, RP, b.q_shared, col_pen, row_pen, s1_end ? 1 : 0, s2_end ? 1 : 0, nb, b.perm, d_out, tbuf, Tmax,
                                   0, b.blockflag, (const int *)nullptr);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) return -(int)e;
                only = b.blockflag;
                g_nwsgv_pt = true;
            }
        }
    }
    if (!only && b.blockflag) {                             // every block bottom-aligned: the walk reads the flags
        const hipError_t e = hipMemsetAsync(b.blockflag, 0xFF, (size_t)blocks * sizeof(int), stream);
        if (e != hipSuccess) return -(int)e;
    }
    hipLaunchKernelGGL((pmx_nwsg16v_kernel<G, R, TR, FETCH, TRB>), dim3((unsigned)blocks), dim3(64), lds, stream,
                       b.qbuf, b.qoff, b.rbuf, b.roff, (long long)b.n, m.scores, m.mapper,
                       m.msize, open, ext, RP, b.q_shared, col_pen, row_pen, s1_end ? 1 : 0, s2_end ? 1 : 0, nb, b.perm, d_out, tbuf, Tmax,
                       0, (int *)nullptr, only);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// Second-generation eligibility: the profile byte score + open must fit, and the proven value range plus
// the skew growth must fit the exact window with the bias chosen here.  Returns the bias, or 0.
int pmx_nwsgv_bias(const PmxBatch &b, const PmxDevMatrix &m, int open, int ext, int rowx, int shape_rows)
{
    if (pmx_env("PMX_NWSG16_GEN1")) return 0;
    if (m.msize > PMX_MAX_FAST_MSIZE - 1 || open < ext || ext < 0 || b.max_rlen > 30000) return 0;
    const long long hi = (long long)(b.max_qlen < b.max_rlen ? b.max_qlen : b.max_rlen) * (m.max > 0 ? m.max : 0) + (m.max > 0 ? m.max : 0);
    long long lo, growth;
    // rows of the shape that will hold the query: at most twice the query's (+ the smallest shapes' 160 / 192), at most 2 048
    // (shape_rows > 0: the launcher's re-check with the G * R of the shape it really picked -- short queries against long references can
    //  land on a larger shape than this estimate, and the kernels' own row offsets and capture bias use the real one)
    const long long rows = shape_rows > 0 ? shape_rows
                         : 2 * b.max_qlen + 64 < 192 ? 192 : 2 * b.max_qlen + 64 < 2048 ? 2 * b.max_qlen + 64 : 2048;
    if (rowx) {
        // Row offset + column skew: a value of cell (i, j) is stored + (i + j) ext, and H(i, j) >= -(2 open + (i + j) ext) always (a
        // gap along row -1, then one down column j): what is stored never falls more than 2 open (+ the one step of E / F / H - C /
        // the diagonal sum below it) under the form of a true 0 -- the decline along the gaps that `lo` covers without the row offset
        // is cancelled by the offsets.  The offsets grow the top instead: + ext per column and per row of the shape (at most twice
        // the query's rows, 2 048).  Scores as bytes: + open + ext.
        lo = -(3LL * open + (m.min < 0 ? -m.min : 0) + 2LL * ext);
        growth = (long long)(b.max_rlen + 2 * 64 + 4) * ext + rows * ext;
        if (m.max + open + ext > 255) return 0;
        // the free-end captures: nb + cb + H in [0, 2^15), cb = 4 open + (rows + rlen + 2) ext (the kernels' own, at most)
        const long long cbmax = 4LL * open + (rows + b.max_rlen + 2) * ext;
        if ((1536 - lo + open) + cbmax + hi + 2LL * open + 64 >= 32767) return 0;
    } else {
        lo = -(3LL * open + (long long)(b.max_qlen + b.max_rlen + 2) * ext + (m.min < 0 ? -m.min : 0));
        growth = (long long)(b.max_rlen + 2 * 64 + 4) * ext;           // (column skew: + ext per column)
    }
    const long long span = (hi - lo) + growth + 2LL * open + (m.max > 0 ? m.max : 0) + 2048;
    if (m.min + open < 0 || m.max + open > 255 || span >= 31743) return 0;
    return (int)(1536 - lo + open);
}

extern "C" int pmx_window_nwsgv(int max_qlen, int max_rlen, int msize, int score_min, int score_max, int open, int ext, int rowx, int shape_rows)
{
    PmxBatch b = {};
    b.max_qlen = max_qlen; b.max_rlen = max_rlen; b.n = 1;
    PmxDevMatrix m = {};
    m.msize = msize; m.min = score_min; m.max = score_max;
    return pmx_nwsgv_bias(b, m, open, ext, rowx, shape_rows);
}

// Traceback variant (global / semi-global): shapes with 16 rows per lane.  Trace layout: per block
// Tmax steps x 64 lanes x 16 bytes.
int pmx_nwsgv_trace_plan(const PmxBatch &b, const PmxDevMatrix &m, int mode, int open, int ext,
                         int *variant, int *Tmax, size_t *trace_bytes)
{
    if (mode != PMX_MODE_NW && mode != PMX_MODE_SG) return 1;
    if (b.q_shared || b.perm || !pmx_nwsgv_bias(b, m, open, ext, 1)) return 1;
    if (m.msize > 8 && m.msize < 32 && !pmx_env("PMX_NWSG16_NO_MATRIX_LOOKUP")) {   // large alphabet: the matrix-lookup kernel (1 KB of LDS)
        int G = 0;
        for (int v = 1; v < 4 && !G; ++v) if (b.max_qlen <= (8 << v) * 16 - 1) { *variant = 4 + v; G = 8 << v; }
        if (!G) return 1;
        *Tmax = (b.max_rlen + G - 1 + 1 + 15) & ~15;    // multiple of 16: the walk reads a lane's records in windows of 16
        *trace_bytes = (size_t)((b.n + 2 * (64 / G) - 1) / (2 * (64 / G))) * (size_t)*Tmax * 64 * 16;
        return 0;
    }
    int G = 0;
    for (int v = 0; v < 4 && !G; ++v) {                  // the first shape that holds the query and fits the LDS (launch_nwsgv's condition)
        const int g = 8 << v, np = 2 * (64 / g);
        const size_t lds = (size_t)np * (m.msize + 1) * g * 16 + (size_t)np * (b.max_rlen + 2 * g + 12) +
                           (size_t)m.msize * m.msize * 2 + 256 + 8 + (size_t)np * 40;
        if (b.max_qlen <= g * 16 - 1 && lds <= 160 * 1024) { *variant = v; G = g; }
    }
    if (!G) return 1;
    const int NP = 2 * (64 / G);
    *Tmax = (b.max_rlen + G - 1 + 1 + 15) & ~15;    // multiple of 16: the walk reads a lane's records in windows of 16
    *trace_bytes = (size_t)((b.n + NP - 1) / NP) * (size_t)*Tmax * 64 * 16;
    return 0;
}

int pmx_launch_nwsgv_trace(int variant, const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                           pmx_record_t *d_out, uint32_t *tbuf, int Tmax, hipStream_t stream)
{
    const int nb = pmx_nwsgv_bias(b, m, open, ext, 1);
    if (!nb) return 1;
    switch (variant) {
    case 5: return launch_nwsgm<16, 16, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    case 6: return launch_nwsgm<32, 16, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    case 7: return launch_nwsgm<64, 16, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    case 0: return launch_nwsgv<8, 16, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    case 1: if (pmx_env("PMX_TRACE_FETCH")) return launch_nwsgv<16, 16, true, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
            return launch_nwsgv<16, 16, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    case 2: return launch_nwsgv<32, 16, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    case 3: return launch_nwsgv<64, 16, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream, tbuf, Tmax);
    }
    return 1;
}

// 0 launched, 1 not eligible (caller uses the general kernel), <0 HIP error
int pmx_launch_nwsg16(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                      pmx_record_t *d_out, hipStream_t stream, const char **kernel_name)
{
    if (pmx_env("PMX_NO_FAST_NWSG")) return 1;
    if (mode != PMX_MODE_NW && mode != PMX_MODE_SG) return 1;
    if (m.msize > PMX_MAX_FAST_MSIZE - 1) return 1;
    if (open < ext || open < 0 || ext < 0) return 1;           // the virtual-row/column fixed points need open >= extend
    if (b.max_rlen > 30000) return 1;
    const int q = b.max_qlen;
    // second-generation arithmetic (skewed columns, byte profile, VOP2) whenever its window holds
    // (second generation, scores only: the row above lane 0 is a closed form, so no virtual row is needed and the
    //  query may fill all G * R rows; the first generation and the traceback walk need row -1 to exist)
    if (const int nb = pmx_nwsgv_bias(b, m, open, ext, b.track8 ? 0 : 1)) {      // (width 8's range tracking: the form without the row offset)
        if (b.q_shared && !b.track8 && !pmx_env("PMX_NWSG16_NO_SHARED")) {        // profile arm: one profile per workgroup, references from HBM
#define TRYQ(GG, RR, NAME)                                                      \
            if (q <= (GG) * (RR)) {                                         \
                int rc = launch_nwsgq<GG, RR>(b, m, mode, sg_flags, open, ext, nb, d_out, stream); \
                if (rc <= 0) { if (kernel_name) *kernel_name = NAME; return rc; } \
            }
            TRYQ(16, 10, "pmx_nwsg16q_kernel<16,10>/shared profile")
            TRYQ(16, 16, "pmx_nwsg16q_kernel<16,16>/shared profile")
            TRYQ(32, 10, "pmx_nwsg16q_kernel<32,10>/shared profile")
            TRYQ(32, 16, "pmx_nwsg16q_kernel<32,16>/shared profile")
            TRYQ(64, 16, "pmx_nwsg16q_kernel<64,16>/shared profile")
            TRYQ(64, 32, "pmx_nwsg16q_kernel<64,32>/shared profile")
#undef TRYQ
        }
        if (!b.q_shared && !b.track8 && m.msize > 8 && m.msize < 32 && b.n > 2048 && !pmx_env("PMX_NWSG16_NO_MATRIX_LOOKUP")) {   // per-pair, large alphabet
#define TRYM(GG, RR, NAME)                                                      \
            if (q <= (GG) * (RR)) {                                             \
                int rc = launch_nwsgm<GG, RR>(b, m, mode, sg_flags, open, ext, nb, d_out, stream); \
                if (rc <= 0) { if (kernel_name) *kernel_name = NAME; return rc; } \
            }
            TRYM(16, 10, "pmx_nwsg16m_kernel<16,10>/matrix lookup")
            TRYM(16, 16, "pmx_nwsg16m_kernel<16,16>/matrix lookup")
            TRYM(32, 10, "pmx_nwsg16m_kernel<32,10>/matrix lookup")
            TRYM(32, 16, "pmx_nwsg16m_kernel<32,16>/matrix lookup")
            TRYM(64, 16, "pmx_nwsg16m_kernel<64,16>/matrix lookup")
#undef TRYM
        }
        const bool longref = b.max_rlen >= 1024 && !pmx_env("PMX_NWSG16_NO_FETCH");   // staged references would dominate the LDS
#define TRYV(GG, RR, NAME)                                                      \
        if (q <= (GG) * (RR)) {                                             \
            int rc = longref ? launch_nwsgv<GG, RR, false, true>(b, m, mode, sg_flags, open, ext, nb, d_out, stream) \
                             : launch_nwsgv<GG, RR>(b, m, mode, sg_flags, open, ext, nb, d_out, stream); \
            if (rc <= 0) { if (kernel_name) *kernel_name = longref ? NAME "/fetch" : g_nwsgv_pt ? NAME "/permtable (+ LDS profiles for marked blocks)" : NAME; return rc; }   \
        }
        if (b.n > 2048) {       // (few pairs: latency counts, the 16-lane shapes have half the work per step)
            TRYV(8, 7, "pmx_nwsg16v_kernel<8,7>")        // reads of 50 / 75 / 100 / 125 / 150 bp: few padding rows
            TRYV(8, 10, "pmx_nwsg16v_kernel<8,10>")
            TRYV(8, 13, "pmx_nwsg16v_kernel<8,13>")
            TRYV(8, 16, "pmx_nwsg16v_kernel<8,16>")
            TRYV(8, 19, "pmx_nwsg16v_kernel<8,19>")
            TRYV(8, 20, "pmx_nwsg16v_kernel<8,20>")
        }
        // one pair or a handful (Aligner::align()): latency = steps x rows per lane of one wave; all 64 lanes on the pair
        // (pmx_sw16.hip has the same ladder)
#define TRYLAT(RR)                                                              \
        if (!longref && b.n <= 64 && q <= 64 * (RR) - 1) {                      \
            const int rc = launch_nwsgv<64, RR>(b, m, mode, sg_flags, open, ext, nb, d_out, stream); \
            if (rc <= 0) { if (kernel_name) *kernel_name = "pmx_nwsg16v_kernel<64," #RR ">"; return rc; } \
        }
        TRYLAT(2) TRYLAT(3) TRYLAT(4) TRYLAT(8)
#undef TRYLAT
        TRYV(16, 10, "pmx_nwsg16v_kernel<16,10>")
        TRYV(16, 16, "pmx_nwsg16v_kernel<16,16>")
        TRYV(32, 10, "pmx_nwsg16v_kernel<32,10>")
        TRYV(32, 16, "pmx_nwsg16v_kernel<32,16>")
        TRYV(64, 16, "pmx_nwsg16v_kernel<64,16>")
        TRYV(64, 32, "pmx_nwsg16v_kernel<64,32>")
#undef TRYV
    }
    if (b.track8) return 1;                                    // (the first-generation kernel does not track the range: general kernel)
    // exact window of the biased lanes: every H, E, F, H-open, E-ext and H(diag)+score stays inside
    const long long lo = -(3LL * open + (long long)(b.max_qlen + b.max_rlen + 2) * ext + (m.min < 0 ? -m.min : 0));
    const long long hi = (long long)(b.max_qlen < b.max_rlen ? b.max_qlen : b.max_rlen) * (m.max > 0 ? m.max : 0) + (m.max > 0 ? m.max : 0);
    if (lo < -15000 || hi > 15000) return 1;
#define TRYN(GG, RR, NAME)                                                      \
    if (q <= (GG) * (RR) - 1) {                                                 \
        int rc = launch_nwsg<GG, RR>(b, m, mode, sg_flags, open, ext, d_out, stream); \
        if (rc <= 0) { if (kernel_name) *kernel_name = NAME; return rc; }       \
    }
    TRYN(16, 10, "pmx_nwsg16_kernel<16,10>")
    TRYN(16, 16, "pmx_nwsg16_kernel<16,16>")
    TRYN(32, 10, "pmx_nwsg16_kernel<32,10>")
    TRYN(32, 16, "pmx_nwsg16_kernel<32,16>")
    TRYN(64, 16, "pmx_nwsg16_kernel<64,16>")
    TRYN(64, 32, "pmx_nwsg16_kernel<64,32>")
#undef TRYN
    return 1;
}
