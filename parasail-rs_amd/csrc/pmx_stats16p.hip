// pmx_stats16p.hip -- second-generation kernel for global / semi-global alignment WITH statistics
// (matches, similar, length): two pairs per lane slot in the int16 halves of every register.
// BASELINE config 3: `nw_stats_striped_profile_16`, one reused query against many references
// (/root/reference/src/aligner/mod.rs:431-450, stats getters src/alignment/mod.rs:79-98).  gfx950 only.
//
// Mapping of pmx_stats16.hip (strip-systolic, query TOP-aligned: lane 0 gets the top boundary
// arithmetically, the G-1 virtual columns in front of the reference reproduce the left boundary, length
// increments are switched off on virtual columns), arithmetic of pmx_nwsg16.hip's second generation:
//   * values of column j are kept as value + nb + (j + G) * ext (no subtract for the E extension), the byte
//     profile carries score + open, add / subtract are 32-bit VOP2 on both halves at once, and a penalised
//     virtual column scores -open instead of -inf (it can never beat the F chain there; needs open >= ext);
//   * the three statistics travel with H, E and F exactly as in the oracle (coupled tables, same tie-breaks:
//     diag, then F, then E; "open" only when strictly greater).  Every decision is the sign of a packed
//     difference (v_pk_sub_i16, v_pk_ashrrev_i16 -> a 0xFFFF / 0 mask per half), every select one v_bfi_b32
//     per statistic: 34 instructions per 2 cells against 2 x 24 in the unpacked kernel;
//   * reference symbols are not staged in LDS (16 pairs of 5 kaa would take 80 KB and the occupancy with them):
//     each lane fetches its next symbols from HBM two steps ahead (neighbouring lanes read neighbouring bytes);
//   * match / similar increments come from two more byte planes of the LDS profile.  With a shared query
//     (profile arm) the planes are built once per 4-wave workgroup.
#include "pmx_common.h"
#include "pmx_switches.h"
#include <cstdlib>

typedef short p_v2s __attribute__((ext_vector_type(2)));
typedef unsigned short p_v2u __attribute__((ext_vector_type(2)));
typedef _Float16 p_v2h __attribute__((ext_vector_type(2)));
#define P_PK(x)  __builtin_bit_cast(p_v2s, (int)(x))
#define P_I32(x) __builtin_bit_cast(int, (x))

__device__ __forceinline__ int p_max3(int a, int b, int c)       // v_pk_maximum3_f16: exact integer max3 on [1024, 31743] patterns
{
    const p_v2h r = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(p_v2h, a), __builtin_bit_cast(p_v2h, b)),
                                                  __builtin_bit_cast(p_v2h, c));
    return __builtin_bit_cast(int, r);
}
__device__ __forceinline__ int p_lt(int a, int b)                // per half: 0xFFFF where a < b (values below 32768)
{
    const p_v2s sh = {15, 15};
    return P_I32((P_PK(a) - P_PK(b)) >> sh);
}
__device__ __forceinline__ int p_bfi(int m, int a, int b)        // (m & a) | (~m & b)
{
    int r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
template <int G>
__device__ __forceinline__ int p_shift_up(int x)                 // value of lane - 1 (group boundaries are fixed up by the caller)
{
    if (G <= 16) return __builtin_amdgcn_update_dpp(x, x, 0x111 /*row_shr:1*/, 0xF, 0xF, false);
    return __builtin_amdgcn_update_dpp(x, x, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
}

struct PCand { int H; int i; int jL; int MS; };                  // true score, row, column | length << 16, matches | similar << 16

// ML (per-pair queries): no profile planes at all -- the score is a byte read from the transposed matrix in LDS (see
// pmx_sw16m.hip), the similar increment is "score > 0" read off that byte, the match increment a packed comparison
// of the query letter codes (kept in registers) with the reference codes of the step.
template <int G, int R, int WAVES, bool ML>
__global__ __launch_bounds__(64 * WAVES)
void pmx_stats16p_kernel(const uint8_t *__restrict__ qbuf, const int64_t *__restrict__ qoff,
                         const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff,
                         long long n, const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap,
                         int msize, int open, int ext, int RP, int q_shared,
                         int col_pen, int row_pen, int s1_end, int s2_end, int nb,
                         const unsigned *__restrict__ perm,
                         pmx_record_t *__restrict__ out, pmx_stats_t *__restrict__ stats_out)
{
    constexpr int RS = (R + 3) / 4 * 4;         // profile bytes reserved per lane (whole dwords)
    constexpr int QPS = G * RS;                 // profile bytes per (pair, symbol, plane)
    constexpr int NPW = 2 * (64 / G);           // pairs per wave
    constexpr int NP = NPW * WAVES;             // pairs per workgroup
    constexpr int NT = 64 * WAVES;
    constexpr int W4 = RS / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane % G, slotw = lane / G;
    const int pA = wave * NPW + 2 * slotw, pB = pA + 1;
    const int MS1 = msize + 1;
    constexpr int MSTR = 32;                    // ML: bytes per row of the transposed matrix
    const int NPROF = ML ? 0 : (q_shared ? 1 : NP);
    const int PLANE = ML ? (MS1 * MSTR + 2) / 3 : NPROF * MS1 * QPS;        // bytes per plane (ML: the three "planes" just hold matT)

    // LDS carve: [score plane][match plane][similar plane][mat][map][ptab]
    unsigned char *psc = lds, *pim = lds + PLANE, *pis = lds + 2 * PLANE;
    int16_t *mat = reinterpret_cast<int16_t *>(lds + ((3 * PLANE + 7) & ~7));
    unsigned char *map = reinterpret_cast<unsigned char *>(mat + msize * msize);
    long long *ptab = reinterpret_cast<long long *>(map + 256 + ((8 - ((msize * msize * 2) & 7)) & 7));

    const long long pair0 = (long long)blockIdx.x * NP;
    for (int i = tid; i < msize * msize; i += NT) mat[i] = gmat[i];
    for (int i = tid; i < 256; i += NT) map[i] = gmap[i];
    if (tid < NP) {
        long long pos = pair0 + tid; if (pos >= n) pos = n - 1;
        const long long pi = perm ? (long long)perm[pos] : pos;
        const long long qb = q_shared ? 0 : qoff[pi], rb = roff[pi];
        ptab[5 * tid + 0] = qb;
        ptab[5 * tid + 1] = q_shared ? q_shared : (qoff[pi + 1] - qb);
        ptab[5 * tid + 2] = rb;
        ptab[5 * tid + 3] = roff[pi + 1] - rb;
        ptab[5 * tid + 4] = (pair0 + tid < n) ? pi : -1;
    }
    __syncthreads();

    // ---- profile planes: query row i = l * R + k sits at byte l * RS + k (rows >= qlen score 0, no increments)
    const int vcol_b = col_pen ? 0 : open;      // real row x pad symbol (virtual column)
    if (ML) {       // matT[r][q]: letters score + open; column msize = rows beyond the query (score 0); row msize = the pad symbol
        for (int i = tid; i < MS1 * MSTR; i += NT) {
            const int r = i / MSTR, q = i % MSTR;
            lds[i] = (unsigned char)(r < msize ? (q < msize ? mat[q * msize + r] + open : open) : (q < msize ? vcol_b : open));
        }
    }
    for (int p = 0; p < NPROF; ++p) {
        const int qlp = (int)ptab[5 * p + 1];
        const uint8_t *qp = qbuf + ptab[5 * p + 0];
        for (int er = tid; er < G * R; er += NT) {
            const int q0 = (er < qlp) ? (int)map[qp[er]] : -1;
            const int pos = (er / R) * RS + er % R;
            unsigned char *sc = psc + (size_t)p * MS1 * QPS + pos, *im = pim + (size_t)p * MS1 * QPS + pos, *is = pis + (size_t)p * MS1 * QPS + pos;
            for (int sym = 0; sym < msize; ++sym) {
                const int s = (q0 < 0) ? 0 : mat[q0 * msize + sym];
                sc[sym * QPS] = (unsigned char)(s + open);
                im[sym * QPS] = (unsigned char)(q0 >= 0 && q0 == sym);
                is[sym * QPS] = (unsigned char)(q0 >= 0 && s > 0);
            }
            sc[msize * QPS] = (unsigned char)((q0 < 0) ? open : vcol_b);
            im[msize * QPS] = 0; is[msize * QPS] = 0;
        }
    }
    __syncthreads();

    // ---- per-lane state ------------------------------------------------------------------
    const int prA = (q_shared || ML) ? 0 : pA, prB = (q_shared || ML) ? 0 : pB;
    const unsigned char *scA = psc + (size_t)prA * MS1 * QPS + g * RS, *scB = psc + (size_t)prB * MS1 * QPS + g * RS;
    const int qlA = (int)ptab[5 * pA + 1], qlB = (int)ptab[5 * pB + 1];
    const int rlA = (int)ptab[5 * pA + 3], rlB = (int)ptab[5 * pB + 3];
    const uint8_t *refA = rbuf + ptab[5 * pA + 2], *refB = rbuf + ptab[5 * pB + 2];
    // symbol of step x for this lane: column x - g of the reference, the pad symbol outside it
    auto fetch = [&](int x, int &ra, int &rb) {
        const int col = x - g;
        ra = (col >= 0 && col < rlA) ? (int)refA[col] : -1;
        rb = (col >= 0 && col < rlB) ? (int)refB[col] : -1;
    };
    auto sym_of = [&](int raw) -> int { return raw < 0 ? msize : (int)map[raw]; };
    // ML: letter codes of this lane's rows (msize beyond the query): LDS offsets inside a matT row, and packed for the match test
    int qa[ML ? R : 1], qb_[ML ? R : 1], qc[ML ? R : 1];
    if (ML) {
        const uint8_t *qA = qbuf + ptab[5 * pA + 0] + g * R, *qB = qbuf + ptab[5 * pB + 0] + g * R;
        unsigned char ra[R], rb[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            ra[k] = g * R + k < qlA ? qA[k] : (unsigned char)0;
            rb[k] = g * R + k < qlB ? qB[k] : (unsigned char)0;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            qa[k] = g * R + k < qlA ? (int)map[ra[k]] : msize;
            qb_[k] = g * R + k < qlB ? (int)map[rb[k]] : msize;
            qc[k] = qa[k] | (qb_[k] << 16);
        }
    }
    auto pack2 = [](int a, int b) -> int { return (a & 0xFFFF) | (b << 16); };
    const int vExt = pack2(ext, ext), vC = pack2(open - ext, open - ext), vOpenP = pack2(open, open);
    const int one2 = 0x00010001;
    const int base = nb + (G - g) * ext - open;      // X-form of a true 0 in column j0 - 1 (j0 = -g is this lane's first column)

    int X[R], E[R], hM[R], hS[R], hL[R], eM[R], eS[R], eL[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int i = g * R + k;
        const int ht = col_pen ? -(open + i * ext) : 0, lt = col_pen ? i + 1 : 0;     // H(i, virtual column) and its length
        X[k] = pack2(base + ht, base + ht); E[k] = X[k];
        hM[k] = hS[k] = 0; hL[k] = pack2(lt, lt);
        eM[k] = eS[k] = 0; eL[k] = pack2(lt + 1, lt + 1);
    }
    int Hout = X[R - 1], HMout = 0, HSout = 0, HLout = hL[R - 1];
    int Fout, FMout = 0, FSout = 0, FLout;
    {
        const int i = (g + 1) * R;                  // first row of the lane below, at a virtual column
        const int ft = col_pen ? -(open + i * ext) : -open;
        Fout = pack2(base + open + ft, base + open + ft);
        FLout = col_pen ? pack2(i + 1, i + 1) : one2;
    }
    int diag0, dM0 = 0, dS0 = 0, dL0;
    if (g == 0) { diag0 = pack2(base, base); dL0 = 0; }
    else {
        const int i = g * R - 1;
        const int ht = col_pen ? -(open + i * ext) : 0;
        diag0 = pack2(base + ht, base + ht); dL0 = col_pen ? pack2(i + 1, i + 1) : 0;
    }
    // top boundary (row above lane 0) at lane 0's column t: penalised -> H = -(open + t ext), length t + 1; free -> 0, 0
    int topX = row_pen ? pack2(nb + (G + 1) * ext - 2 * open, nb + (G + 1) * ext - 2 * open)
                       : pack2(nb + (G + 1) * ext - open, nb + (G + 1) * ext - open);
    const int topStep = row_pen ? 0 : vExt;
    int topL = row_pen ? one2 : 0;
    const int topLStep = row_pen ? one2 : 0;

    const int gLA = (qlA - 1) / R, kLA = (qlA - 1) % R, gLB = (qlB - 1) / R, kLB = (qlB - 1) % R;
    PCand corner[2], brow[2], bcol[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { corner[h].H = brow[h].H = bcol[h].H = -(1 << 30); corner[h].i = brow[h].i = bcol[h].i = 0;
                                  corner[h].jL = brow[h].jL = bcol[h].jL = 0; corner[h].MS = brow[h].MS = bcol[h].MS = 0; }

    int wsc[2][2][W4], wim[2][2][W4], wis[2][2][W4];     // [buffer][pair half][dword]
    int wml[2][ML ? R : 1], rc2[2] = {0, 0};             // ML: packed scores of a step, packed reference codes of a step
    auto load_planes = [&](int bsel, int symA, int symB) {
        if (ML) {
            rc2[bsel] = symA | (symB << 16);
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int sa = lds[symA * MSTR + qa[k]], sb = lds[symB * MSTR + qb_[k]];
                wml[bsel][k] = sa | (sb << 16);
            }
            return;
        }
        const int *a0 = reinterpret_cast<const int *>(scA + symA * QPS), *b0 = reinterpret_cast<const int *>(scB + symB * QPS);
        const int *a1 = reinterpret_cast<const int *>(scA + PLANE + symA * QPS), *b1 = reinterpret_cast<const int *>(scB + PLANE + symB * QPS);
        const int *a2 = reinterpret_cast<const int *>(scA + 2 * PLANE + symA * QPS), *b2 = reinterpret_cast<const int *>(scB + 2 * PLANE + symB * QPS);
#pragma unroll
        for (int x = 0; x < W4; ++x) {
            wsc[bsel][0][x] = a0[x]; wsc[bsel][1][x] = b0[x];
            wim[bsel][0][x] = a1[x]; wim[bsel][1][x] = b1[x];
            wis[bsel][0][x] = a2[x]; wis[bsel][1][x] = b2[x];
        }
    };

    auto step = [&](int bsel, int t) {
        const int jcol = t - g;
        // length increments on real columns; on a penalised virtual column too: there the diagonal (score -open) can
        // tie with the F chain when open == ext, and then has to carry the same length (row index + 1)
        const int linc = jcol < 0 ? (col_pen ? one2 : 0)
                                  : ((jcol < rlA ? 1 : 0) | (jcol < rlB ? 0x10000 : 0));
        int Hin = p_shift_up<G>(Hout), HMin = p_shift_up<G>(HMout), HSin = p_shift_up<G>(HSout), HLin = p_shift_up<G>(HLout);
        int F = p_shift_up<G>(Fout), fM = p_shift_up<G>(FMout), fS = p_shift_up<G>(FSout), fL = p_shift_up<G>(FLout);
        if (g == 0) {                                 // top boundary of column t (F^ into row 0 = X of the row above)
            Hin = topX; HMin = 0; HSin = 0; HLin = topL;
            F = topX; fM = 0; fS = 0; fL = topL + one2;
        }
        int T[R], TM[R], TS[R], TL[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const unsigned selw = 0x0C000C00u | (unsigned)(k & 3) | ((4u + (unsigned)(k & 3)) << 16);
            int s, im, is;
            if (ML) {
                const p_v2u fifteen = {15, 15}, one_ = {1, 1};
                s = wml[bsel][k];
                is = P_I32(__builtin_bit_cast(p_v2s, __builtin_bit_cast(p_v2u, P_PK(vOpenP) - P_PK(s)) >> fifteen));            // score + open > open
                im = P_I32(__builtin_bit_cast(p_v2s, (__builtin_bit_cast(p_v2u, qc[k] ^ rc2[bsel]) - one_) >> fifteen));         // equal letter codes
            } else {
                s = __builtin_amdgcn_perm(wsc[bsel][1][k / 4], wsc[bsel][0][k / 4], selw);
                im = __builtin_amdgcn_perm(wim[bsel][1][k / 4], wim[bsel][0][k / 4], selw);
                is = __builtin_amdgcn_perm(wis[bsel][1][k / 4], wis[bsel][0][k / 4], selw);
            }
            T[k] = ((k == 0) ? diag0 : X[k - 1]) + s;
            TM[k] = ((k == 0) ? dM0 : hM[k - 1]) + im;
            TS[k] = ((k == 0) ? dS0 : hS[k - 1]) + is;
            TL[k] = ((k == 0) ? dL0 : hL[k - 1]) + linc;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int Fe = F - vExt;
            const int H = p_max3(T[k], E[k], Fe);
            const int Xn = H - vC;
            const int mNDL = p_lt(Fe, H);                              // not from F -> E's statistics, else F's
            const int xM = p_bfi(mNDL, eM[k], fM), xS = p_bfi(mNDL, eS[k], fS), xL = p_bfi(mNDL, eL[k], fL);
            const int mND = p_lt(T[k], H);                             // not diagonal -> gap statistics, else the diagonal's
            const int nM = p_bfi(mND, xM, TM[k]), nS = p_bfi(mND, xS, TS[k]), nL = p_bfi(mND, xL, TL[k]);
            const int mEO = p_lt(E[k], Xn);                            // E of the next column opened from H
            eM[k] = p_bfi(mEO, nM, eM[k]); eS[k] = p_bfi(mEO, nS, eS[k]); eL[k] = p_bfi(mEO, nL, eL[k]) + one2;
            const int mFO = p_lt(Fe, Xn);                              // F of the next row opened from H
            fM = p_bfi(mFO, nM, fM); fS = p_bfi(mFO, nS, fS); fL = p_bfi(mFO, nL, fL) + one2;
            E[k] = p_max3(E[k], Xn, Xn);
            F = p_max3(Fe, Xn, Xn);
            X[k] = Xn; hM[k] = nM; hS[k] = nS; hL[k] = nL;
        }
        diag0 = Hin; dM0 = HMin; dS0 = HSin; dL0 = HLin;
        Hout = X[R - 1]; HMout = hM[R - 1]; HSout = hS[R - 1]; HLout = hL[R - 1];
        Fout = F; FMout = fM; FSout = fS; FLout = fL;
        topX += topStep; topL += topLStep;

        // ---- captures (per pair half; rare for global alignment: only around each pair's last column) ----
        const bool lastA = jcol == rlA - 1, lastB = jcol == rlB - 1;
        if (s2_end || __builtin_amdgcn_ballot_w64(lastA || lastB) != 0) {
            const int unsk = nb + (jcol + G) * ext - open + ext;       // X-form of a true 0 in this column
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rl = h ? rlB : rlA, ql = h ? qlB : qlA, gL = h ? gLB : gLA, kL = h ? kLB : kLA;
                auto half = [&](int v) -> int { return h ? (int)((unsigned)v >> 16) : (v & 0xFFFF); };
                if (jcol >= 0 && jcol < rl) {
                    if (g == gL && (s2_end || jcol == rl - 1)) {
                        int xv = 0, m_ = 0, s_ = 0, l_ = 0;
#pragma unroll
                        for (int k = 0; k < R; ++k) if (k == kL) { xv = half(X[k]); m_ = half(hM[k]); s_ = half(hS[k]); l_ = half(hL[k]); }
                        const int hv = xv - unsk;
                        PCand c; c.H = hv; c.i = ql - 1; c.jL = jcol | (l_ << 16); c.MS = m_ | (s_ << 16);
                        if (jcol == rl - 1) corner[h] = c;
                        if (s2_end && hv > brow[h].H) brow[h] = c;
                    }
                    if (s1_end && jcol == rl - 1) {
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            const int i = g * R + k, hv = half(X[k]) - unsk;
                            if (i < ql && hv > bcol[h].H) { bcol[h].H = hv; bcol[h].i = i; bcol[h].jL = jcol | (half(hL[k]) << 16); bcol[h].MS = half(hM[k]) | (half(hS[k]) << 16); }
                        }
                    }
                }
            }
        }
    };

    int max_rlen = 0;
#pragma unroll
    for (int p = 0; p < NPW; ++p) max_rlen = max(max_rlen, (int)ptab[5 * (wave * NPW + p) + 3]);
    const int T_ = (max_rlen + G - 1 + 1) & ~1;
    // pipeline: raw bytes two steps ahead (HBM / L2 latency), mapped symbols one step ahead, profile words for the next step
    int r0a, r0b, r1a, r1b, r2a, r2b, r3a, r3b;
    fetch(0, r0a, r0b); fetch(1, r1a, r1b); fetch(2, r2a, r2b); fetch(3, r3a, r3b);
    load_planes(0, sym_of(r0a), sym_of(r0b));
    int nsA = sym_of(r1a), nsB = sym_of(r1b);          // symbols of step t + 1
    int m2a = r2a, m2b = r2b, m3a = r3a, m3b = r3b;    // raw bytes of steps t + 2, t + 3
    for (int t = 0; t < T_; t += 2) {
        load_planes(1, nsA, nsB);
        nsA = sym_of(m2a); nsB = sym_of(m2b);           // step t + 2
        fetch(t + 4, m2a, m2b);
        __builtin_amdgcn_sched_barrier(0);
        step(0, t);
        __builtin_amdgcn_sched_barrier(0);
        load_planes(0, nsA, nsB);
        nsA = sym_of(m3a); nsB = sym_of(m3b);           // step t + 3
        fetch(t + 5, m3a, m3b);
        __builtin_amdgcn_sched_barrier(0);
        step(1, t + 1);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- combine per pair half: last-column candidates over the slot (value desc, row asc), then the oracle's rule ----
    const unsigned long long slotmask = (G == 64) ? ~0ULL : (((1ULL << G) - 1ULL) << (slotw * G));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int gL = h ? gLB : gLA;
        const unsigned key = ((unsigned)(bcol[h].H < -16000 ? 0 : bcol[h].H + 16384) << 16) | (0xFFFFu - (unsigned)bcol[h].i);
        unsigned best = key;
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
            const unsigned o = __shfl_xor(best, off, 64);
            best = o > best ? o : best;
        }
        const unsigned long long win = __ballot(best == key && bcol[h].H > -(1 << 29));
        const int wl = (win & slotmask) ? __builtin_ctzll(win & slotmask) : slotw * G;
        PCand bc, co, br;
        bc.H = __shfl(bcol[h].H, wl, 64); bc.i = __shfl(bcol[h].i, wl, 64); bc.jL = __shfl(bcol[h].jL, wl, 64); bc.MS = __shfl(bcol[h].MS, wl, 64);
        const int ll = slotw * G + gL;
        co.H = __shfl(corner[h].H, ll, 64); co.i = __shfl(corner[h].i, ll, 64); co.jL = __shfl(corner[h].jL, ll, 64); co.MS = __shfl(corner[h].MS, ll, 64);
        br.H = __shfl(brow[h].H, ll, 64); br.i = __shfl(brow[h].i, ll, 64); br.jL = __shfl(brow[h].jL, ll, 64); br.MS = __shfl(brow[h].MS, ll, 64);
        if (g == 0) {
            const long long pi = ptab[5 * ((h ? pB : pA)) + 4];
            if (pi >= 0) {
                PCand res;
                if (!s1_end && !s2_end) res = co;
                else {
                    res.H = -(1 << 30); res.i = 0; res.jL = 0; res.MS = 0;
                    if (s2_end) res = br;
                    if (s1_end && bc.H > res.H) res = bc;
                }
                pmx_record_t rec; rec.score = res.H; rec.end_query = res.i; rec.end_ref = res.jL & 0xFFFF; rec.flags = 0;
                out[pi] = rec;
                pmx_stats_t st; st.matches = res.MS & 0xFFFF; st.similar = (int)((unsigned)res.MS >> 16); st.length = (int)((unsigned)res.jL >> 16);
                stats_out[pi] = st;
            }
        }
    }
}

// ------------------------------------------------------------------------ host side ----
template <int G, int R, int WAVES, bool ML = false>
static int launch_statsp(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext, int nb,
                         pmx_record_t *d_out, pmx_stats_t *d_stats, hipStream_t stream)
{
    constexpr int RS = (R + 3) / 4 * 4, NP = 2 * (64 / G) * WAVES;
    const int RP = ((b.max_rlen + 2 * (G - 1) + 4 + 7) / 4) * 4;
    const int nprof = b.q_shared ? 1 : NP;
    const size_t lds = (ML ? (size_t)(m.msize + 1) * 32 + 8 : (size_t)3 * nprof * (m.msize + 1) * G * RS) + 8 +
                       (size_t)m.msize * m.msize * 2 + 256 + 8 + (size_t)NP * 40;
    if (lds > 160 * 1024) return 1;
    { const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_stats16p_kernel<G, R, WAVES, ML>)); if (rc) return rc; }
    const bool sg = mode == PMX_MODE_SG;
    const int col_pen = !(sg && (sg_flags & PMX_SG_QB)), row_pen = !(sg && (sg_flags & PMX_SG_DB));
    const int s1_end = sg && (sg_flags & PMX_SG_QE), s2_end = sg && (sg_flags & PMX_SG_DE);
    const long long blocks = (b.n + NP - 1) / NP;
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL((pmx_stats16p_kernel<G, R, WAVES, ML>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, stream,
                       b.qbuf, b.qoff, b.rbuf, b.roff, (long long)b.n, m.scores, m.mapper,
                       m.msize, open, ext, RP, b.q_shared, col_pen, row_pen, s1_end ? 1 : 0, s2_end ? 1 : 0, nb,
                       b.perm, d_out, d_stats);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// 0 launched, 1 not eligible (caller tries pmx_stats16), <0 HIP error
int pmx_launch_stats16p(const PmxBatch &b, const PmxDevMatrix &m, int mode, int sg_flags, int open, int ext,
                        pmx_record_t *d_out, pmx_stats_t *d_stats, hipStream_t stream, const char **kernel_name)
{
    if (pmx_env("PMX_NO_FAST_STATS") || pmx_env("PMX_STATS16_GEN1")) return 1;
    if (mode != PMX_MODE_NW && mode != PMX_MODE_SG) return 1;
    // Where it pays (measured): one shared query profile per workgroup (the profile arm, config 3) and results that
    // are captured once per pair (no free end: nw, sg_qb, sg_db, sg_qb_db).  Per-pair profiles cost too much LDS
    // in three planes, and per-step last-row / last-column captures are cheaper in the unpacked kernel.
    // Per-pair queries: only over large alphabets, with the matrix-lookup variant (no profile planes).
    const bool ml = !b.q_shared && m.msize > 8 && m.msize < 32 && !pmx_env("PMX_STATS16P_NO_MATRIX_LOOKUP");
    if (!b.q_shared && !ml && !pmx_env("PMX_STATS16P_ALWAYS")) return 1;
    if (mode == PMX_MODE_SG && (sg_flags & (PMX_SG_QE | PMX_SG_DE)) && !pmx_env("PMX_STATS16P_ALWAYS")) return 1;
    if (ext < 1 || b.max_qlen + b.max_rlen + 2 > 32767) return 1;          // statistics live in int16 halves
    const int nb = pmx_nwsgv_bias(b, m, open, ext);                         // same window proof as the score kernel
    if (!nb) return 1;
    const int q = b.max_qlen;
    const int W = b.q_shared ? 4 : 1;      // a shared query profile is built once per 4-wave workgroup
#define TRYP(GG, RR, NAME)                                                      \
    if (q <= (GG) * (RR)) {                                                     \
        int rc = ml ? launch_statsp<GG, RR, 1, true>(b, m, mode, sg_flags, open, ext, nb, d_out, d_stats, stream)  \
               : W == 4 ? launch_statsp<GG, RR, 4>(b, m, mode, sg_flags, open, ext, nb, d_out, d_stats, stream)  \
                        : launch_statsp<GG, RR, 1>(b, m, mode, sg_flags, open, ext, nb, d_out, d_stats, stream); \
        if (rc <= 0) { if (kernel_name) *kernel_name = ml ? NAME "/matrix lookup" : NAME; return rc; }       \
    }
    TRYP(16, 10, "pmx_stats16p_kernel<16,10>")
    TRYP(32, 10, "pmx_stats16p_kernel<32,10>")
    TRYP(64, 10, "pmx_stats16p_kernel<64,10>")
#undef TRYP
    return 1;
}
