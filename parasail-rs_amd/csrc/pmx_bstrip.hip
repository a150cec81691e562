// pmx_bstrip.hip -- banded alignment in BAND COORDINATES, packed int16, two pairs per lane group.  gfx950 only.
//
// Reference counterpart: Aligner::banded_nw -> parasail_nw_banded (/root/reference/src/aligner/mod.rs:454-489; KAT
// tests/test_parasail.rs:726-736) and the batch extension of include/parasail_amd.h (any mode, per-pair band centre: BASELINE
// config 5's "banded SW").  The band rule is the oracle's (oracle/pmx_oracle.c:orc_align_ex): cell (i, j) belongs to the band iff
// |(j - i) - diag| <= band; outside it H = E = F = -inf; the boundary row / column keep their values.
//
// Why another band kernel: pmx_banded.hip gives a lane ONE cell per step (anti-diagonal wavefront), so every cell pays the whole
// per-step overhead -- 23.8 VALU instructions per 128 band cells where the strip kernels of the full matrix pay 9.9
// (profiles/r03/cfg5_banded_pmc_summary.json); on reads a band of 31 was slower than no band.  Here a lane owns C consecutive band
// OFFSETS d = j - i - (diag - band) and walks the query one row per step, so the recurrences turn by a quarter:
//
//     T(i, d) = H(i-1, d)   + S(q_i, r_j)            same offset, previous row: the lane's own register
//     E(i, d) = max(E(i, d-1) - ext, H(i, d-1) - open)   along the row: a chain through the lane's C cells, then to lane g + 1
//     F(i, d) = max(F(i-1, d+1) - ext, H(i-1, d+1) - open)   previous row, next offset: the lane's own register, the last cell's from lane g + 1
//
// Lane g + 1 needs lane g's E of the SAME row, lane g needs lane g + 1's F of the PREVIOUS row: a row of the group takes two slots per
// lane.  Every lane splits its cells in a left and a right half and alternates between them; the halves form a wavefront of 2 G
// units with one DPP move per slot, and no lane ever waits (lane g works on row u - g in wave step u).  Only band cells are computed:
// no triangles at the ends of a strip, no masks in the loop.
//
// Arithmetic (the model, cell for cell: tests/bstrip_model.py, checked against the banded oracle on the CPU tier):
//   * values are STORED as true + sigma(tau, d) + bias in the window [1024, 31743] where v_pk_maximum3_f16 is an exact integer max3
//     (profiles/microbench/max3_f16_int.hip); sigma = ext * (tau + d) for global / semi-global (E along the row needs no subtraction),
//     ext * (2 tau + d) in the double-skew variant (F needs none either: 6 instructions per two cells), ext * tau for local alignment
//     (the zero floor is ONE value per row and rides on the E chain);
//   * scores come from v_perm_b32: the table is the 4 score bytes of the row's query letter (one dword per pair from LDS), the
//     selector is the cell's reference letter; the selectors slide one cell per row (a shift of the register names every 4 rows);
//   * columns left of the matrix are VIRTUAL: selector 0x0C (score byte 0) keeps them low, and for a free query begin the column
//     just left of the matrix gets selector 0x0D (byte 255) on top of a level the band's edge input holds the virtual cells at, so the
//     ordinary recurrence lands exactly on the boundary value; a penalised boundary column is the F chain's closed form; the
//     boundary ROW is the initial state.  The band's edges are inputs (a per-row E at the first offset, one F at the last);
//   * offsets in front of the band (the lanes' capacity G * C exceeds 2 band + 1) are GUARDED -- never updated, so they keep their
//     initial "low" and pass the edge input through;
//   * a reference letter beyond the first four (a wildcard has no selector) hands the pair back through the retry list: the
//     launcher runs those pairs in pmx_banded_kernel.
#include "pmx_common.h"
#include "pmx_switches.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>

typedef short s_v2s __attribute__((ext_vector_type(2)));
typedef _Float16 s_v2h __attribute__((ext_vector_type(2)));
#define SPK(x) __builtin_bit_cast(s_v2s, (int)(x))
#define SI32(x) __builtin_bit_cast(int, (x))
#define S_NEG (INT32_MIN / 2)

__device__ __forceinline__ int s_max3(int a, int b, int c)      // exact integer max3 on patterns in [1024, 31743]
{
    const s_v2h r = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(s_v2h, a), __builtin_bit_cast(s_v2h, b)),
                                                  __builtin_bit_cast(s_v2h, c));
    return __builtin_bit_cast(int, r);
}
__device__ __forceinline__ int s_max2(int a, int b) { return SI32(__builtin_elementwise_max(SPK(a), SPK(b))); }

// value of group member g - 1 (lanes of a group sit 16 / G apart inside a DPP row of 16); member 0 keeps `old`
template <int G>
__device__ __forceinline__ int s_from_prev(int x, int old)
{
    if (G == 1) return old;
    return __builtin_amdgcn_update_dpp(old, x, 0x110 + (G == 1 ? 1 : 16 / G) /* row_shr */, 0xF, 0xF, false);
}
// value of group member g + 1; the last member keeps `old`
template <int G>
__device__ __forceinline__ int s_from_next(int x, int old)
{
    if (G == 1) return old;
    return __builtin_amdgcn_update_dpp(old, x, 0x100 + (G == 1 ? 1 : 16 / G) /* row_shl */, 0xF, 0xF, false);
}

struct SGeo {                 // one pair of a lane group (one int16 half)
    long long qb, rb, pair;
    int ql, rl, j0, i_s, rows, have, miss;
};

struct SConst {
    int mode, sg_flags, open, ext, band, bias, low;
};

// Initial state of offset d (model: init_of): Hx = X form of H one row above the first, Fn = max(F, X) of that row.
template <int MODEV>
__device__ __forceinline__ void s_init_of(const SGeo &p, const SConst &k, int d, bool col_pen, bool row_pen, int &hx, int &fn)
{
    constexpr bool SW = MODEV >= 2;
    constexpr int A_ = SW ? 1 : (MODEV == 1 ? 2 : 1), B_ = SW ? 0 : 1;
    const int Cg = k.open - k.ext;
    const int sig = (A_ * -1 + B_ * d) * k.ext;
    hx = k.low; fn = k.low;
    if (p.miss || d < 0) return;
    if (SW) { hx = k.bias + sig - Cg; return; }
    const int jp = p.i_s + p.j0 + d - 1;
    if (jp <= -2) {
        if (!col_pen) {
            // K_E(-1) - Cg: the level of the virtual cells, 255 below the boundary column's stored value of the first row
            const int d_b = -1 - p.i_s - p.j0;                       // offset of column -1 in the first row
            hx = (B_ * d_b) * k.ext + k.bias - 255;                  // target(0) - 255 + Cg - Cg   (tau = 0: alpha * 0)
        }
        return;
    }
    if (p.i_s == 0) {
        const int tv = jp == -1 ? 0 : (row_pen ? -(k.open + jp * k.ext) : 0);
        hx = tv + sig + k.bias - Cg; fn = hx;
        return;
    }
    if (jp != -1) return;
    const int tv = col_pen ? -(k.open + (p.i_s - 1) * k.ext) : 0;
    hx = tv + sig + k.bias - Cg;
    fn = col_pen ? hx + Cg : hx;
}

template <int G, int C, int MODEV /* 0: nw / sg, one skew; 1: nw / sg, double skew; 2: sw; 3: sw, ties settled by a second launch */,
          int EPG /* > 0: that many leading cells carry a guard of their own; < 0: the first -EPG cells are ONE guarded block (the band starts at cell -EPG of the first live lane) */>
__global__ __launch_bounds__(64)
void pmx_bstrip_kernel(const uint8_t *__restrict__ qbuf, const int64_t *__restrict__ qoff, int q_shared,
                       const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff, long long n,
                       const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap, int msize,
                       SConst k, const int32_t *__restrict__ diag, const unsigned *__restrict__ perm,
                       int RC, int QC /* bytes per lane group: reference selectors, query letters (one shared query: QC bytes once) */,
                       unsigned *__restrict__ retry_list, int *__restrict__ retry_count,
                       unsigned *__restrict__ tie_list, int *__restrict__ tie_count /* MODEV 3: pairs whose end cell a tie may move */,
                       const int *__restrict__ n_dev /* != nullptr: the pair count is read from the device (the launch over tie_list) */,
                       pmx_record_t *__restrict__ out)
{
    constexpr bool SW = MODEV >= 2, TIES_LATER = MODEV == 3;
    if (n_dev) n = *n_dev;
    if ((long long)blockIdx.x * (2 * (64 / G)) >= n) return;
    constexpr int A_ = SW ? 1 : (MODEV == 1 ? 2 : 1), B_ = SW ? 0 : 1;      // sigma = (A_ tau + B_ d) ext
    constexpr bool ESUB = B_ == 0, FSUB = A_ == B_;
    constexpr int SUBG = 16 / G, NG = 64 / G, NP = 2 * NG;
    constexpr int CL = (C + 1) / 2;                  // cells of the left half
    constexpr int U = 4;                             // rows between two shifts of the selector registers
    constexpr int PADF = G - 1;                      // query stream: entries in front of the first row (lanes that have not started)

    __shared__ unsigned char map[256];
    __shared__ uint2 tab2[36];                       // (letter a of pair A, letter b of pair B) -> the two score dwords; 5 = no row
    __shared__ int gwild[NP];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];   // [NG][RC] reference selectors, [NG][QC] query letters ([QC]: one shared query)

    const int lane = threadIdx.x;
    const int l16 = lane & 15, g = l16 / SUBG, sub = l16 % SUBG, grp = (lane >> 4) * SUBG + sub;
    const bool sg = k.mode == PMX_MODE_SG;
    const bool s1_end = sg && (k.sg_flags & PMX_SG_QE), s2_end = sg && (k.sg_flags & PMX_SG_DE);
    const bool col_pen = k.mode == PMX_MODE_NW || (sg && !(k.sg_flags & PMX_SG_QB));   // H(i, -1) penalised
    const bool row_pen = k.mode == PMX_MODE_NW || (sg && !(k.sg_flags & PMX_SG_DB));   // H(-1, j) penalised
    const int W = 2 * k.band + 1, eL = G * C - W, gLo = eL / C, cLoFirst = eL % C;
    const int open = k.open, ext = k.ext, Cg = open - ext;
    const int OB = Cg + A_ * ext;                    // score byte = score + OB

    // ---- tables ----
    for (int x = lane; x < 256; x += 64) map[x] = gmap[x];
    if (lane < 36) {
        const int a = lane % 6, b = lane / 6;
        unsigned ta = 0, tb = 0;
        for (int c = 0; c < 4 && c < msize; ++c) {
            if (a < msize) ta |= (unsigned)((gmat[a * msize + c] + OB) & 0xFF) << (8 * c);
            if (b < msize) tb |= (unsigned)((gmat[b * msize + c] + OB) & 0xFF) << (8 * c);
        }
        tab2[lane] = make_uint2(ta, tb);
    }
    if (lane < NP) gwild[lane] = 0;

    // ---- geometry of the group's two pairs ----
    SGeo P[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long long pos = (long long)blockIdx.x * NP + 2 * grp + h;
        SGeo &p = P[h];
        p.have = pos < n;
        const long long pp = perm ? (long long)perm[p.have ? pos : n - 1] : (p.have ? pos : n - 1);
        p.pair = pp;
        p.qb = q_shared ? 0 : qoff[pp]; p.rb = roff[pp];
        p.ql = q_shared ? q_shared : (int)(qoff[pp + 1] - p.qb); p.rl = (int)(roff[pp + 1] - p.rb);
        const int d0 = diag ? diag[pp] : 0;
        p.j0 = d0 - k.band;
        p.miss = (p.j0 > p.rl - 1) || (d0 + k.band < -(p.ql - 1));
        p.i_s = max(0, -p.j0 - W + 1);
        const int i_e = min(p.ql - 1, p.rl - 1 - p.j0);
        p.rows = i_e - p.i_s + 1;
        if (p.miss || p.rows <= 0) { p.miss = 1; p.rows = 0; p.i_s = 0; p.j0 = p.rl + 256; }
    }
    __syncthreads();

    // ---- staging: per lane group ONE byte stream of reference selectors (pair A in the low nibble, pair B in the high one:
    //      both pairs read position t = u + g (C - 1) + c of their own windows) and one of query letters (a + 6 b).
    //      The G lanes of a group stage their own streams, SB positions per batch: all loads of a batch are in flight before
    //      anything consumes them (the prologue is latency-bound otherwise) ----
    unsigned char *rs_all = dyn, *qs_all = dyn + (size_t)NG * RC;
    {
        constexpr int SB = 12;
        unsigned char *rsg = rs_all + (size_t)grp * RC, *qsg = qs_all + (size_t)grp * QC;
        const int jbA = P[0].i_s + P[0].j0 - eL, jbB = P[1].i_s + P[1].j0 - eL;       // column of stream position 0
        const int edge = (SW || col_pen) ? 0xC : 0xD;
        const uint8_t *rA = rbuf + P[0].rb, *rB = rbuf + P[1].rb, *qA = qbuf + P[0].qb, *qB = qbuf + P[1].qb;
        int wa = 0, wb = 0;
        for (int t0 = g; t0 < RC; t0 += G * SB) {
            int ra[SB], rb[SB];
#pragma unroll
            for (int x = 0; x < SB; ++x) {
                const int t = t0 + x * G, jA = jbA + t, jB = jbB + t;
                ra[x] = (jA >= 0 && jA < P[0].rl) ? (int)rA[jA] : -1;
                rb[x] = (jB >= 0 && jB < P[1].rl) ? (int)rB[jB] : -1;
            }
#pragma unroll
            for (int x = 0; x < SB; ++x) {
                const int t = t0 + x * G, jA = jbA + t, jB = jbB + t;
                int ca = jA == -1 ? edge : 0xC, cb = jB == -1 ? edge : 0xC;
                if (ra[x] >= 0) { const int L = map[ra[x]]; if (L < 4) ca = L; else wa = 1; }
                if (rb[x] >= 0) { const int L = map[rb[x]]; if (L < 4) cb = 4 + L; else wb = 1; }
                if (t < RC) rsg[t] = (unsigned char)(ca | (cb << 4));
            }
        }
        if (q_shared) {
            // one shared query: its letters once per wave (index = row + PADF), every pair reads from its own first row on
            for (int x0 = lane; x0 < QC; x0 += 64 * SB) {
                int qa[SB];
#pragma unroll
                for (int y = 0; y < SB; ++y) { const int x = x0 + y * 64, i = x - PADF; qa[y] = (i >= 0 && i < q_shared) ? (int)qbuf[i] : -1; }
#pragma unroll
                for (int y = 0; y < SB; ++y) {
                    const int x = x0 + y * 64;
                    int a = 5;
                    if (qa[y] >= 0) { a = map[qa[y]]; if (a > 4) a = 4; }
                    if (x < QC) qs_all[x] = (unsigned char)a;
                }
            }
        } else
        for (int x0 = g; x0 < QC; x0 += G * SB) {
            int qa[SB], qb[SB];
#pragma unroll
            for (int y = 0; y < SB; ++y) {
                const int x = x0 + y * G, iA = P[0].i_s + x - PADF, iB = P[1].i_s + x - PADF;
                qa[y] = (x >= PADF && iA < P[0].ql) ? (int)qA[iA] : -1;
                qb[y] = (x >= PADF && iB < P[1].ql) ? (int)qB[iB] : -1;
            }
#pragma unroll
            for (int y = 0; y < SB; ++y) {
                const int x = x0 + y * G;
                int a = 5, b = 5;
                if (qa[y] >= 0) { a = map[qa[y]]; if (a > 4) a = 4; }
                if (qb[y] >= 0) { b = map[qb[y]]; if (b > 4) b = 4; }
                if (x < QC) qsg[x] = (unsigned char)(a + 6 * b);
            }
        }
        if (wa) gwild[2 * grp] = 1;
        if (wb) gwild[2 * grp + 1] = 1;
    }
    __syncthreads();

    // ---- per lane: initial state ----
    const int ext2 = ext * 0x00010001, Cg2 = Cg * 0x00010001, LOW2 = k.low * 0x00010001;
    int Hx[C], Fn[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int d = g * C + c - eL;
        int hA, fA, hB, fB;
        s_init_of<MODEV>(P[0], k, d, col_pen, row_pen, hA, fA);
        s_init_of<MODEV>(P[1], k, d, col_pen, row_pen, hB, fB);
        Hx[c] = (hA & 0xFFFF) | (hB << 16); Fn[c] = (fA & 0xFFFF) | (fB << 16);
    }
    int Fedge;                                       // F input of the band's last offset: the boundary row in the first row, nothing after
    {
        int hA, fA, hB, fB;
        s_init_of<MODEV>(P[0], k, W, col_pen, row_pen, hA, fA);
        s_init_of<MODEV>(P[1], k, W, col_pen, row_pen, hB, fB);
        Fedge = (fA & 0xFFFF) | (fB << 16);
    }
    const int rowsLane = max(P[0].rows, P[1].rows);
    int nsteps = rowsLane ? rowsLane + G - 1 : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) nsteps = max(nsteps, __shfl_xor(nsteps, off, 64));
    nsteps = (nsteps + U - 1) / U * U;
    const int cLo = g == gLo ? cLoFirst : 0;         // cells in front of the band in the first live lane: guarded
    const bool laneLive = g >= gLo;

    // edge input of E (first live lane; model: "E entering the band's first cell"); per half:
    //   jL = i + j0 <= -1: the level of the virtual cells (free query begin) or LOW;  jL == 0: E opened from the boundary column;
    //   jL >= 1: the band's edge lies inside the matrix: LOW
    int tau0[2], e0[2], ke0[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const SGeo &p = P[h];
        tau0[h] = p.miss ? -1 : -(p.i_s + p.j0);                     // row (relative) whose first band cell is column 0
        const int i0 = p.i_s + tau0[h];                             // = -j0
        const int bc = col_pen ? -(open + i0 * ext) : 0;
        e0[h] = bc + (A_ * tau0[h] - B_) * ext + k.bias + B_ * ext - open;        // target(tau0) + beta - open
        // K_E(tau) = target(tau + 1) - 255 + Cg, free boundary: (A_ (tau + 1) + B_ d_b(tau + 1)) ext + bias - 255 + Cg, d_b(t) = -1 - i_s - t - j0
        ke0[h] = (A_ + B_ * (-2 - p.i_s - p.j0)) * ext + k.bias - 255 + Cg;       // at tau = 0; grows by (A_ - B_) ext per row
    }
    int uE = -1;                                     // last wave step in which some first live lane needs a scheduled edge input
    if (!SW && g == gLo) uE = max(tau0[0], tau0[1]) + g;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) uE = max(uE, __shfl_xor(uE, off, 64));

    // ---- streams ----
    const unsigned char *rs = rs_all + (size_t)grp * RC + g * (C - 1);        // position of (u, c): rs[u + c]
    const unsigned char *qs = qs_all + (size_t)grp * QC + (PADF - g);          // row of wave step u: qs[u]
    const unsigned char *qsA = qs_all + P[0].i_s + (PADF - g), *qsB = qs_all + P[1].i_s + (PADF - g);   // one shared query: the pairs' own rows
    int S[C + U];
    auto sel_of = [&](int byte) -> int {             // (A nibble, B nibble) -> {A, 0x0C, B, 0x0C}
        return (((byte << 12) | byte) & 0x000F000F) | 0x0C000C00;
    };
#pragma unroll
    for (int x = 0; x < C; ++x) S[x] = sel_of(rs[x]);
#pragma unroll
    for (int x = C; x < C + U; ++x) S[x] = 0x0C0C0C0C;

    int Eout = LOW2;
    int Zpe = SW ? (k.bias + ext) * 0x00010001 : 0;  // sw: stored zero of the lane's row + ext
    const int a2 = A_ * ext * 0x00010001;

    // captures
    int corner[2] = {S_NEG, S_NEG};
    int browH[2] = {S_NEG, S_NEG}, browJ[2] = {0, 0}, bcolH[2] = {S_NEG, S_NEG}, bcolI[2] = {0, 0};
    int best = SW ? (k.bias - Cg) * 0x00010001 : 0;  // sw: the lane's best in the X form of its current row
    int bestrow = 0;                                 // sw: the row (relative, per half) that saved it
    int tflag = 0;                                   // sw, ties settled later: bit 15 of a half = a tie with the lane's own best since it was saved
    int fake = -1;                                   // sw: halves whose `best` is not a score this lane has seen (the initial zero; a bound taken over from the group)
    int Hsave[SW ? C : 1];                           // sw: that row's strip
#pragma unroll
    for (int c = 0; c < (SW ? C : 1); ++c) Hsave[c] = 0;
    const int tauM[2] = {(!P[0].miss && P[0].i_s + P[0].rows == P[0].ql) ? P[0].ql - 1 - P[0].i_s : -0x40000000,
                         (!P[1].miss && P[1].i_s + P[1].rows == P[1].ql) ? P[1].ql - 1 - P[1].i_s : -0x40000000};
    // last column: the lane's cell index of column rl - 1 in row tau is lc0 - tau
    const int lc0[2] = {P[0].rl - 1 - P[0].i_s - P[0].j0 - (g * C - eL), P[1].rl - 1 - P[1].i_s - P[1].j0 - (g * C - eL)};

    // (one shared query: tab2[a].x and tab2[6 b].y are the single letters' tables)
    // rows in which this lane may have something to capture (corner / last row: tauM; last column: C rows ending at lc0): one range
    int evLo = 0x7FFFFFFF, evHi = -1;
    if (!SW) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (tauM[h] >= 0) { evLo = min(evLo, tauM[h]); evHi = max(evHi, tauM[h]); }
            if (s1_end && !P[h].miss && lc0[h] >= 0 && lc0[h] - (C - 1) < P[h].rows) { evLo = min(evLo, max(0, lc0[h] - (C - 1))); evHi = max(evHi, lc0[h]); }
        }
    }
    const unsigned evSpan = evHi >= evLo ? (unsigned)(evHi - evLo) : 0u;
    if (evHi < evLo) evLo = 0x7FFFFFFF;
    uint2 tabN = q_shared ? make_uint2(tab2[qsA[0]].x, tab2[6 * qsB[0]].y) : tab2[qs[0]];
    int qbN = q_shared ? (int)qsA[1] : (int)qs[1], qbN2 = q_shared ? (int)qsB[1] : 0;      // (one shared query: the two letters, combined when their tables are read)
    int rsN = rs[C];

    for (int u0 = 0; u0 < nsteps; u0 += U) {
#pragma unroll
        for (int rho = 0; rho < U; ++rho) {
            const int u = u0 + rho, tau = u - g;
            // ---- loads of the next row: its tables (the letter byte was read a row ago), the letter byte after that, the selector
            //      that enters at the lane's last cell ----
            const uint2 tab = tabN;
            tabN = tab2[qbN + 6 * qbN2];
            if (q_shared) { qbN = qsA[u + 2]; qbN2 = qsB[u + 2]; } else qbN = qs[u + 2];
            S[C + rho] = sel_of(rsN);                // belongs to row u + 1's last cell
            rsN = rs[u + C + 1];
            __builtin_amdgcn_sched_barrier(0);
            const bool active = laneLive && tau >= 0 && tau < rowsLane;

            // ---- edge input of E ----
            int sched = LOW2;
            if (SW) sched = Zpe;
            else if (u <= uE) {
                int e[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    e[h] = k.low;
                    if (tau == tau0[h]) e[h] = e0[h];
                    else if (tau < tau0[h] && !col_pen) e[h] = ke0[h] + (A_ - B_) * ext * tau;
                }
                sched = (e[0] & 0xFFFF) | (e[1] << 16);
            }
            int Ein = s_from_prev<G>(Eout, sched);
            if (gLo > 0) Ein = g == gLo ? sched : Ein;

            int E = Ein;
            auto cell = [&](int c, int Fin) {
                const int s = (int)__builtin_amdgcn_perm(tab.y, tab.x, (unsigned)S[c + rho]);
                const int T = Hx[c] + s;
                const int Fe = FSUB ? Fin - ext2 : Fin;
                const int Ee = ESUB ? E - ext2 : E;
                const int H = s_max3(T, Ee, Fe);
                const int X = H - Cg2;
                E = SW ? s_max3(Ee, X, Zpe) : s_max2(Ee, X);
                Fn[c] = s_max2(Fe, X);
                Hx[c] = X;
            };
            // ---- left half: E arrives from lane g - 1 (its right half of the same row, one slot ago) ----
            if (EPG < 0) {
                static_assert(EPG >= 0 || -EPG <= CL, "a guarded block lies inside the left half");
                if (active && g != gLo) {
#pragma unroll
                    for (int c = 0; c < -EPG && c < CL; ++c) cell(c, Fn[c + 1]);
                }
                if (active) {
#pragma unroll
                    for (int c = (EPG < 0 ? -EPG : 0); c < CL; ++c) cell(c, Fn[c + 1]);
                }
            } else if (active) {
#pragma unroll
                for (int c = 0; c < CL; ++c) {
                    if (c < EPG) { if (c >= cLo) cell(c, Fn[c + 1]); }
                    else cell(c, Fn[c + 1]);
                }
            }
            // ---- right half: F of the last cell arrives from lane g + 1 (its left half of the previous row, this slot) ----
            const int FinL = s_from_next<G>(Fn[0], Fedge);
            if (active) {
#pragma unroll
                for (int c = CL; c < C; ++c) {
                    const int Fin = c + 1 < C ? Fn[c + 1] : FinL;
                    if (c < EPG) { if (c >= cLo) cell(c, Fin); }
                    else cell(c, Fin);
                }
                Eout = E;
                Fedge = LOW2;
                if (SW) {
                    // ---- local: the lane's best, exact in column-major order (smallest column, then smallest row).
                    //      A row that strictly exceeds the lane's best saves the row index and the strip (v_bfi under a per-half mask;
                    //      related pairs improve in almost every row, so this is the common path); a row that only EQUALS it can still
                    //      win by an earlier column (j = tau + c: an earlier cell index by more than the rows in between): rare, exact ----
                    int rm = Hx[0];
#pragma unroll
                    for (int c = 1; c + 1 < C; c += 2) rm = s_max3(rm, Hx[c], Hx[c + 1]);
                    if ((C & 1) == 0) rm = s_max2(rm, Hx[C - 1]);
                    const s_v2s fifteen = {15, 15};
                    const int d2 = SI32(SPK(best) - SPK(rm));
                    const int mgt = SI32(SPK(d2) >> fifteen);                        // 0xFFFF per half where the row exceeds the best
                    // a half with d2 == 0 whose best is one this lane has SEEN (a bound taken over from the group cannot be tied: it lies
                    // below the group's best)
                    // ... and only while the saved cell is young: a later row wins a tie by an earlier COLUMN (tau + c), i.e. within C - 2 rows
                    const int age = (tau & 0xFFFF) * 0x00010001 - bestrow;
                    const int tie = (d2 - 0x00010001) & ~d2 & (int)0x80008000 & ~fake & (age - (C - 1) * 0x00010001);
                    // TIES_LATER: such a tie is only REMEMBERED (until the lane's next real improvement); a pair whose winning lane ends with one is
                    // handed to a second launch of the exact variant -- related reads tie in most rows SOMEWHERE in a wave, and settling each on the
                    // spot cost a quarter of the kernel
                    if (TIES_LATER) tflag |= tie;
                    else if (__builtin_amdgcn_ballot_w64(tie != 0) != 0) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int r16 = h ? (int)((unsigned)rm >> 16) : (rm & 0xFFFF);
                            if (r16 != (h ? (int)((unsigned)best >> 16) : (best & 0xFFFF)) || (h ? (fake >> 16) : (fake & 0xFFFF))) continue;   // (the packed test may flag a half next to a zero one)
                            const int srow = h ? (int)((unsigned)bestrow >> 16) : (bestrow & 0xFFFF);
                            const int sbest = r16 - A_ * ext * (tau - srow);         // the best in the form of the row that saved it
                            int c2 = C, c1 = C;
#pragma unroll
                            for (int c = C - 1; c >= 0; --c) {
                                const int x16 = h ? (int)((unsigned)Hx[c] >> 16) : (Hx[c] & 0xFFFF);
                                const int v16 = h ? (int)((unsigned)Hsave[c] >> 16) : (Hsave[c] & 0xFFFF);
                                if (x16 == r16) c2 = c;
                                if (v16 == sbest) c1 = c;
                            }
                            if (tau + c2 < srow + c1) {
                                const int m = h ? (int)0xFFFF0000 : 0x0000FFFF;
                                bestrow = (bestrow & ~m) | (((tau & 0xFFFF) * 0x00010001) & m);
#pragma unroll
                                for (int c = 0; c < C; ++c) Hsave[c] = (Hsave[c] & ~m) | (Hx[c] & m);
                            }
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(mgt != 0) != 0) {
                        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(bestrow) : "v"(mgt), "v"((tau & 0xFFFF) * 0x00010001), "v"(bestrow));
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            int hs;
                            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hs) : "v"(mgt), "v"(Hx[c]), "v"(Hsave[c]));
                            Hsave[c] = hs;
                        }
                        best = s_max2(best, rm);
                        fake &= ~mgt;
                        if (TIES_LATER) tflag &= ~mgt;
                    }
                    best += a2;
                    Zpe += a2;
                } else {
                    // ---- global / semi-global: the corner, the last row, the last column ----
                    if (__builtin_amdgcn_ballot_w64((unsigned)(tau - evLo) <= evSpan) != 0) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const SGeo &p = P[h];
                            const int i = p.i_s + tau;
                            if (p.miss || tau >= p.rows) continue;
                            const int lc = lc0[h] - tau;
                            const bool lastrow = tau == tauM[h];
                            const bool incol = s1_end && (unsigned)lc < (unsigned)C && g * C + lc - eL >= 0;
                            if (incol) {                                             // the cell of column rl - 1: one select per cell
                                int x16 = 0;
#pragma unroll
                                for (int c = 0; c < C; ++c) x16 = c == lc ? (h ? (int)((unsigned)Hx[c] >> 16) : (Hx[c] & 0xFFFF)) : x16;
                                const int tv = x16 + Cg - k.bias - (A_ * tau + B_ * (g * C + lc - eL)) * ext;
                                if (tv > bcolH[h]) { bcolH[h] = tv; bcolI[h] = i; }
                            }
                            if (!lastrow) continue;
#pragma unroll
                            for (int c = 0; c < C; ++c) {
                                const int d = g * C + c - eL, j = i + p.j0 + d;
                                if (d < 0 || j < 0 || j >= p.rl) continue;
                                const int x16 = h ? (int)((unsigned)Hx[c] >> 16) : (Hx[c] & 0xFFFF);
                                const int tv = x16 + Cg - k.bias - (A_ * tau + B_ * d) * ext;
                                if (j == p.rl - 1) corner[h] = tv;
                                if (s2_end && tv > browH[h]) { browH[h] = tv; browJ[h] = j; }
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int x = 0; x < C; ++x) S[x] = S[x + U];
        if (SW && G > 1 && (u0 & 15) == 12) {
            // Every 16 steps the lanes of a group agree on a lower bound of their pairs' scores -- the largest best any of them holds, in the
            // form of row 0 -- and each raises its own best to one BELOW it: a lane whose rows stay under the bound can no longer hold the
            // end cell, so its improvements and ties (the lanes off the alignment's diagonal tie with their small bests in almost every
            // row) stop costing anything, while a lane that reaches the bound itself still saves.  A best raised this way is not a score the
            // lane has seen (`fake`); the epilogue gives such halves no candidate.
            const int rowsRun = max(0, min(u0 + U - g, rowsLane));                   // `best` is in the form of that row
            const int form = ((A_ * ext * rowsRun) & 0xFFFF) * 0x00010001;
            int v = laneLive ? best - form : 0;
#pragma unroll
            for (int off = G / 2; off >= 1; off >>= 1) v = s_max2(v, __shfl_xor(v, SUBG * off, 64));
            const int fresh = v - 0x00010001 + form;
            const int nb = s_max2(best, fresh);
            const s_v2s fifteen = {15, 15};
            fake |= SI32((SPK(best) - SPK(nb)) >> fifteen);                          // halves that were raised
            best = nb;
        }
    }

    // ---- reduction over the group, records ----
    int bestT[2] = {0, 0}, bestJ[2] = {0x7FFFFFFF, 0x7FFFFFFF}, bestI[2] = {0, 0};     // sw: true score, column, row of the lane's best
    if (SW) {
        const int rowsDone = max(0, min(nsteps - g, rowsLane));                     // rows this lane ran: `best` was rebased once per row
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int b16 = h ? (int)((unsigned)best >> 16) : (best & 0xFFFF);
            const int srow = h ? (int)((unsigned)bestrow >> 16) : (bestrow & 0xFFFF);
            const int sbest = b16 - A_ * ext * (rowsDone - srow);
            int c1 = C;
#pragma unroll
            for (int c = C - 1; c >= 0; --c) {
                const int v16 = h ? (int)((unsigned)Hsave[c] >> 16) : (Hsave[c] & 0xFFFF);
                if (v16 == sbest) c1 = c;
            }
            if (c1 < C && laneLive && !(h ? (fake >> 16) : (fake & 0xFFFF))) {
                bestT[h] = sbest + Cg - k.bias - A_ * ext * srow;
                bestI[h] = P[h].i_s + srow;
                bestJ[h] = P[h].i_s + srow + P[h].j0 + g * C + c1 - eL;
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const SGeo &p = P[h];
        int cr = corner[h], bh = browH[h], bj = browJ[h], ch = bcolH[h], ci = bcolI[h], st = bestT[h], sj = bestJ[h], si = bestI[h];
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
            const int lo = SUBG * off;
            cr = max(cr, __shfl_xor(cr, lo, 64));
            { const int oh = __shfl_xor(bh, lo, 64), oj = __shfl_xor(bj, lo, 64); if (oh > bh || (oh == bh && oj < bj)) { bh = oh; bj = oj; } }
            { const int oh = __shfl_xor(ch, lo, 64), oi = __shfl_xor(ci, lo, 64); if (oh > ch || (oh == ch && oi < ci)) { ch = oh; ci = oi; } }
            { const int ot = __shfl_xor(st, lo, 64), oj = __shfl_xor(sj, lo, 64), oi = __shfl_xor(si, lo, 64);
              if (ot > st || (ot == st && (oj < sj || (oj == sj && oi < si)))) { st = ot; sj = oj; si = oi; } }
        }
        int unsure = 0;
        if (TIES_LATER) {
            // (the packed tie test may flag a half next to a zero one: then a pair is redone for nothing, never the other way round)
            unsure = (bestT[h] == st && st > 0 && ((h ? (tflag >> 16) : tflag) & 0x8000)) ? 1 : 0;
#pragma unroll
            for (int off = G / 2; off >= 1; off >>= 1) unsure |= __shfl_xor(unsure, SUBG * off, 64);
        }
        if (g == 0 && p.have) {
            pmx_record_t rec; rec.flags = 0;
            if (SW) {
                if (p.miss) { rec.score = S_NEG; rec.end_query = 0; rec.end_ref = 0; }
                else if (st <= 0) {
                    // no cell above 0: the band's first cell in column-major order
                    const int dlo = p.j0, dhi = p.j0 + 2 * k.band;
                    const int j = max(0, dlo), i = max(0, j - dhi);
                    rec.score = 0; rec.end_query = i; rec.end_ref = j;
                } else { rec.score = st; rec.end_query = si; rec.end_ref = sj; }
            } else if (k.mode == PMX_MODE_NW || (!s1_end && !s2_end)) {
                rec.score = p.miss ? S_NEG : cr; rec.end_query = p.ql - 1; rec.end_ref = p.rl - 1;
            } else {
                int rh = S_NEG, ri = 0, rj = 0;
                if (!p.miss) {
                    if (bh > S_NEG) { rh = bh; ri = p.ql - 1; rj = bj; }
                    if (s1_end && ch > rh) { rh = ch; ri = ci; rj = p.rl - 1; }
                }
                rec.score = rh; rec.end_query = ri; rec.end_ref = rj;
            }
            if (gwild[2 * grp + h]) {
                rec.flags = PMX_FLAG_RETRY16;
                retry_list[atomicAdd(retry_count, 1)] = (unsigned)p.pair;
            } else if (TIES_LATER && unsure && !p.miss) {
                rec.flags = PMX_FLAG_RETRY16;
                tie_list[atomicAdd(tie_count, 1)] = (unsigned)p.pair;
            }
            out[p.pair] = rec;
        }
    }
}

// ------------------------------------------------------------------------ host side ----

// The window predicate (model: bias_and_low in tests/bstrip_model.py, compared entry by entry on the CPU tier through the
// exported hook below): the bias B of the stored form and the LOW constant, or false when the int16 window cannot hold a launch
// with queries <= m rows, references <= n columns and `rows` row steps per pair at most.
static bool bstrip_window(int mode, int m, int n, int open, int ext, int smin, int smax, int cap, int rows, bool double_skew,
                          int *bias, int *low)
{
    const int a = mode == PMX_MODE_SW ? 1 : (double_skew ? 2 : 1), b = mode == PMX_MODE_SW ? 0 : 1;
    const long long Cg = (long long)open - ext;
    if (open < ext || ext < 0 || Cg > 120) return false;
    const long long OB = Cg + (long long)a * ext;
    if (smin + OB < 0 || smax + OB > 254) return false;
    const long long LOW = 1024 + 2LL * std::max(open, ext) + 8;
    const long long L = std::min(m, n);
    long long lo_true, hi_true = (long long)std::max(0, smax) * L;
    if (mode == PMX_MODE_SW) lo_true = 0;
    else lo_true = -((long long)open + (long long)std::max(m, n) * ext) + (long long)std::min(0, smin) * L - open;
    const long long sig_hi = ((long long)a * (rows + 2) + (long long)b * (cap + 2)) * ext;
    const long long sig_lo = -(long long)(a + b) * ext * 2;
    const long long need_lo = LOW + 2LL * open + 300 + (long long)rows * ext;
    const long long B = need_lo - (lo_true + sig_lo);
    const long long top = hi_true + sig_hi + B + 256;
    if (top > 31743 - 8) return false;
    *bias = (int)B; *low = (int)LOW;
    return true;
}
extern "C" int pmx_bstrip_window(int mode, int m, int n, int open, int ext, int smin, int smax, int cap, int rows, int double_skew,
                                 int *bias, int *low)
{
    int b = 0, l = 0;
    const bool ok = bstrip_window(mode, m, n, open, ext, smin, smax, cap, rows, double_skew != 0, &b, &l);
    if (bias) *bias = b;
    if (low) *low = l;
    return ok ? 1 : 0;
}

// lanes per pair and offsets per lane for a band: the smallest capacity G * C >= 2 band + 1
struct BsShape { int G, C; };
// (capacity 32 on four lanes: <2,16> keeps 32 lane groups' streams in LDS -- two waves per SIMD -- and measured 2.17 ms against
//  1.77 for <4,8> on 1.25 M reads of 250 x 250 with band 15)
static const BsShape bs_shapes[] = {{1, 8}, {1, 12}, {1, 16}, {2, 12}, {4, 8}, {4, 12}, {4, 16}, {8, 12}, {8, 13}, {8, 16}};
static const BsShape bs_more[] = {{2, 16}, {8, 8}};        // reachable through PMX_BSTRIP_SHAPE only
extern "C" int pmx_bstrip_shape(int band, int *G, int *C)
{
    if (const char *f = pmx_env("PMX_BSTRIP_SHAPE")) {         // "GxC": that shape if it exists and holds the band
        int fg = 0, fc = 0;
        if (sscanf(f, "%dx%d", &fg, &fc) == 2 && fg * fc >= 2 * band + 1) {
            for (const BsShape &s : bs_shapes) if (s.G == fg && s.C == fc) { if (G) *G = fg; if (C) *C = fc; return 1; }
            for (const BsShape &s : bs_more) if (s.G == fg && s.C == fc) { if (G) *G = fg; if (C) *C = fc; return 1; }
        }
    }
    for (const BsShape &s : bs_shapes)
        if (s.G * s.C >= 2 * band + 1) { if (G) *G = s.G; if (C) *C = s.C; return 1; }
    return 0;
}

void pmx_banded_retry(int mode, int sg_flags, int open, int ext, const PmxDevMatrix &m, long long n,
                      const uint8_t *qbuf, const int64_t *qoff, int q_shared, const uint8_t *rbuf, const int64_t *roff,
                      int band, const int32_t *diag, const unsigned *list, const int *count, pmx_record_t *out, hipStream_t stream);

template <int G, int C, int MODEV>
static int bs_launch(int guard /* 0: one leading cell, 1: every cell, 2: the first 7 cells of <8,13> as one block */, unsigned blocks, const int *n_dev, long long n, const uint8_t *qbuf, const int64_t *qoff, int q_shared, const uint8_t *rbuf, const int64_t *roff,
                     const PmxDevMatrix &m, const SConst &k, const int32_t *diag, const unsigned *perm, int RC, int QC, size_t lds,
                     unsigned *retry_list, int *retry_count, pmx_record_t *out, hipStream_t stream)
{
#define BS_GO(EPG) do { \
        if (lds > 48 * 1024) { const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_bstrip_kernel<G, C, MODEV, EPG>), 150 * 1024); if (rc) return rc; } \
        hipLaunchKernelGGL((pmx_bstrip_kernel<G, C, MODEV, EPG>), dim3(blocks), dim3(64), lds, stream, qbuf, qoff, q_shared, rbuf, roff, n, \
                           m.scores, m.mapper, m.msize, k, diag, perm, RC, QC, retry_list, retry_count, \
                           retry_list + n + 1, reinterpret_cast<int *>(retry_list + n), n_dev, out); } while (0)
    if (guard == 2) { if constexpr (G == 8 && C == 13) BS_GO(-7); else return 1; }
    else if (guard == 1) BS_GO(C); else BS_GO(1);
#undef BS_GO
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// 0 launched (retry pairs included), 1 not eligible, < 0 HIP error
int pmx_launch_bstrip(int mode, int sg_flags, int open, int ext, const PmxDevMatrix &m, long long n,
                      const uint8_t *qbuf, const int64_t *qoff, int q_shared, const uint8_t *rbuf, const int64_t *roff,
                      int max_qlen, int max_rlen, int band, const int32_t *diag, pmx_record_t *out, hipStream_t stream,
                      const char **kernel_name, void *sort_scratch, unsigned *retry_list, int *retry_count)
{
    if (pmx_env("PMX_BANDED_NO_STRIP") || !retry_list || !retry_count) return 1;
    if (m.msize > 5 || n <= 0 || n >= (1LL << 31)) return 1;
    int G = 0, C = 0;
    if (!pmx_bstrip_shape(band, &G, &C)) return 1;
    const bool sw = mode == PMX_MODE_SW;
    const bool ds = !sw && !pmx_env("PMX_BSTRIP_ONE_SKEW");
    const int cap = G * C, rows = std::min(max_qlen, max_rlen + band + 1);
    SConst k;
    k.mode = mode; k.sg_flags = sg_flags; k.open = open; k.ext = ext; k.band = band;
    int modev = sw ? 2 : (ds ? 1 : 0);
    if (!bstrip_window(mode, max_qlen, max_rlen, open, ext, m.min, m.max, cap, rows, modev == 1, &k.bias, &k.low)) {
        if (modev != 1 || !bstrip_window(mode, max_qlen, max_rlen, open, ext, m.min, m.max, cap, rows, false, &k.bias, &k.low)) return 1;
        modev = 0;
    }
    const int U = 4;
    // (one shared query: a pair whose band enters the matrix at row i_s reads letters up to i_s + the wave's step count)
    const int QC = ((q_shared ? 2 * max_qlen : rows) + 2 * G + U + 8 + 3) & ~3, RC = (rows + G + U + cap + 8 + 3) & ~3;
    const int NG = 64 / G;
    const size_t lds = (size_t)NG * RC + (q_shared ? (size_t)QC : (size_t)NG * QC);
    if (lds > 148 * 1024) return 1;
    const int eL = cap - (2 * band + 1);
    // offsets in front of the band in the first live lane: one (the common case: capacity = band width + 1) has a guard of its own; <8,13>
    // with band 48 (7 cells = its whole left half: config 5's second pass) skips them as one block; anything else guards every cell
    const int guard_all = (eL % C) <= 1 ? 0 : (G == 8 && C == 13 && eL == 7 && !pmx_env("PMX_BSTRIP_CELL_GUARDS")) ? 2 : 1;
    // scratch: [retry_count][retry_list: n][tie_count][tie_list: n]
    int *tie_count = reinterpret_cast<int *>(retry_list + n);
    hipError_t e = hipMemsetAsync(retry_count, 0, sizeof(int), stream);
    if (e == hipSuccess) e = hipMemsetAsync(tie_count, 0, sizeof(int), stream);
    if (e != hipSuccess) return -(int)e;
    const unsigned *perm = nullptr;
    if (sort_scratch && n >= 4096) {
        const int rc = pmx_build_band_perm(qoff, q_shared, roff, diag, band, n, sort_scratch, &perm, stream, false);
        if (rc < 0) return rc;
    }
    // local alignment: the first launch only REMEMBERS ties that could move an end cell (variant 3); the few pairs it is unsure of
    // are redone by the exact variant (2), driven by the device-side count
    const bool ties_later = modev == 2 && !pmx_env("PMX_BSTRIP_TIES_INLINE");
    const unsigned blocks = (unsigned)((n + 2 * (64 / G) - 1) / (2 * (64 / G)));
    int rc = 1;
#define BS_SHAPE(GG, CC) if (G == GG && C == CC) { \
        rc = modev == 2 ? (ties_later ? bs_launch<GG, CC, 3>(guard_all, blocks, nullptr, n, qbuf, qoff, q_shared, rbuf, roff, m, k, diag, perm, RC, QC, lds, retry_list, retry_count, out, stream) \
                                      : bs_launch<GG, CC, 2>(guard_all, blocks, nullptr, n, qbuf, qoff, q_shared, rbuf, roff, m, k, diag, perm, RC, QC, lds, retry_list, retry_count, out, stream)) \
           : modev == 1 ? bs_launch<GG, CC, 1>(guard_all, blocks, nullptr, n, qbuf, qoff, q_shared, rbuf, roff, m, k, diag, perm, RC, QC, lds, retry_list, retry_count, out, stream) \
                        : bs_launch<GG, CC, 0>(guard_all, blocks, nullptr, n, qbuf, qoff, q_shared, rbuf, roff, m, k, diag, perm, RC, QC, lds, retry_list, retry_count, out, stream); \
        if (rc == 0 && ties_later) \
            rc = bs_launch<GG, CC, 2>(guard_all, blocks, tie_count, n, qbuf, qoff, q_shared, rbuf, roff, m, k, diag, retry_list + n + 1, RC, QC, lds, retry_list, retry_count, out, stream); }
    BS_SHAPE(1, 8) BS_SHAPE(1, 12) BS_SHAPE(1, 16) BS_SHAPE(2, 12) BS_SHAPE(2, 16) BS_SHAPE(4, 12) BS_SHAPE(4, 16) BS_SHAPE(8, 12) BS_SHAPE(8, 13) BS_SHAPE(8, 16) BS_SHAPE(4, 8) BS_SHAPE(8, 8)
#undef BS_SHAPE
    if (rc) return rc;
    // pairs with a reference letter beyond the first four: the band-only kernel of pmx_banded.hip, driven by the device-side count
    pmx_banded_retry(mode, sg_flags, open, ext, m, n, qbuf, qoff, q_shared, rbuf, roff, band, diag, retry_list, retry_count, out, stream);
    if (kernel_name) *kernel_name = modev == 2 ? "pmx_bstrip_kernel/local" : modev == 1 ? "pmx_bstrip_kernel/double skew" : "pmx_bstrip_kernel/one skew";
    e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
