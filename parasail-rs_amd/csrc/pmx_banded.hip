// pmx_banded.hip -- banded alignment that computes ONLY the band's cells.  gfx950 only.
//
// Reference counterpart: Aligner::banded_nw -> parasail_nw_banded(s1, s2, open, gap, k, matrix)
// (/root/reference/src/aligner/mod.rs:454-489, "for aligning large sequences"; KAT tests/test_parasail.rs:726-736).
// Extension (BASELINE config 5 "banded SW", no reference counterpart): any mode, and a per-pair band centre --
// cell (i, j) belongs to the band iff |(j - i) - diag| <= band.  The rule is stated once in oracle/pmx_oracle.c
// (orc_align_ex) and restated here.
//
// Mapping: anti-diagonal wavefront inside the band.  A group of LP lanes owns one pair; on step s (= i + j) lane x holds
// the band diagonal u = 2x + p(s) (u = (j - i) - diag + band, 0 <= u <= 2 band; the parity p alternates with s), i.e. at
// most band + 1 cells per step, none outside the band: work is O(qlen * band) instead of O(qlen * rlen).  Neighbours:
// the diagonal predecessor (i-1, j-1) is the lane's own cell two steps ago; on even-u steps the cell above (i-1, j) is
// the lane's own previous cell and the cell to the left (i, j-1) comes from lane x-1, on odd-u steps it is the other
// way round (above from lane x+1) -- one DPP shift of (H, E) or (H, F) per step, no scan along the row.
// 32-bit lanes (no saturation inside a band), score + end positions with the oracle's rules for every mode.
#include "pmx_common.h"
#include "pmx_switches.h"
#include <algorithm>

#define B_NEG (INT32_MIN / 2)

struct BCand { int H, i, j; };
__device__ __forceinline__ bool b_better_sw(const BCand &a, const BCand &b)     // larger H, then smaller j, then smaller i
{
    if (a.H != b.H) return a.H > b.H;
    if (a.j != b.j) return a.j < b.j;
    return a.i < b.i;
}

template <int LP>
__device__ __forceinline__ int b_from_below(int x)      // value of lane - 1 (garbage at the group's first lane: the caller overrides)
{
    if (LP == 16) return __builtin_amdgcn_update_dpp(x, x, 0x111 /*row_shr:1*/, 0xF, 0xF, false);
    return __builtin_amdgcn_update_dpp(x, x, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
}
template <int LP>
__device__ __forceinline__ int b_from_above(int x)      // value of lane + 1
{
    if (LP == 16) return __builtin_amdgcn_update_dpp(x, x, 0x101 /*row_shl:1*/, 0xF, 0xF, false);
    return __builtin_amdgcn_update_dpp(x, x, 0x130 /*wave_shl:1*/, 0xF, 0xF, false);
}

template <int LP>
__global__ __launch_bounds__(64)
void pmx_banded_kernel(const uint8_t *__restrict__ qbuf, const int64_t *__restrict__ qoff, int q_shared,
                       const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff, long long n,
                       const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap, int msize,
                       int mode, int sg_flags, int open, int ext, int band, const int32_t *__restrict__ diag,
                       const unsigned *__restrict__ list /* optional: the pairs to work on ... */, const int *__restrict__ n_dev /* ... and their count (device) */,
                       pmx_record_t *__restrict__ out)
{
    __shared__ int16_t mat[PMX_MAX_FAST_MSIZE * PMX_MAX_FAST_MSIZE];
    __shared__ unsigned char map[256];
    for (int x = threadIdx.x; x < msize * msize; x += 64) mat[x] = gmat[x];
    for (int x = threadIdx.x; x < 256; x += 64) map[x] = gmap[x];
    __syncthreads();

    constexpr int NPW = 64 / LP;
    const int lane = threadIdx.x, x = lane % LP;
    // list form (pairs handed back by pmx_bstrip_kernel): a fixed grid walks the device-side list; otherwise one pass
    const long long nitems = list ? (long long)*n_dev : n;
    for (long long base = (long long)blockIdx.x * NPW; base < nitems; base += (long long)gridDim.x * NPW) {
    const long long item = base + lane / LP;
    const bool have = item < nitems;
    const long long pair = list ? (long long)list[have ? item : nitems - 1] : item;
    const long long pp = have ? pair : (list ? pair : n - 1);
    const long long qb = q_shared ? 0 : qoff[pp], rb = roff[pp];
    const int ql = q_shared ? q_shared : (int)(qoff[pp + 1] - qb), rl = (int)(roff[pp + 1] - rb);
    const uint8_t *q = qbuf + qb, *r = rbuf + rb;
    const int d0 = diag ? diag[pp] : 0;
    const bool sw = mode == PMX_MODE_SW, sg = mode == PMX_MODE_SG;
    const bool s1_end = sg && (sg_flags & PMX_SG_QE), s2_end = sg && (sg_flags & PMX_SG_DE);
    const bool col_pen = mode == PMX_MODE_NW || (sg && !(sg_flags & PMX_SG_QB));   // H(i, -1) penalised
    const bool row_pen = mode == PMX_MODE_NW || (sg && !(sg_flags & PMX_SG_DB));   // H(-1, j) penalised
    auto rowB = [&](int j) -> int { return sw ? 0 : (row_pen ? -(open + j * ext) : 0); };
    auto colB = [&](int i) -> int { return sw ? 0 : (col_pen ? -(open + i * ext) : 0); };

    // step range of this pair: s = i + j over the band's cells inside the matrix
    const int dlo = d0 - band, dhi = d0 + band;                 // j - i ranges over [dlo, dhi]
    int s_first = 0;
    if (dlo > 0) s_first = dlo; else if (dhi < 0) s_first = -dhi;
    // last cell: the largest i + j with i < ql, j < rl, dlo <= j - i <= dhi
    int s_last = -1;
    {
        int i1 = ql - 1, j1 = rl - 1;
        if (j1 - i1 > dhi) j1 = i1 + dhi; else if (j1 - i1 < dlo) i1 = j1 - dlo;
        if (i1 >= 0 && j1 >= 0 && have) s_last = i1 + j1;
    }
    if (dlo > rl - 1 || dhi < -(ql - 1)) s_last = -1;           // the band misses the matrix
    // all groups of the wave run the same number of steps
    int nsteps = s_last - s_first + 1; if (nsteps < 0) nsteps = 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) nsteps = max(nsteps, __shfl_xor(nsteps, off, 64));

    int Hm1 = B_NEG, Em1 = B_NEG, Fm1 = B_NEG, Hm2 = B_NEG;
    BCand best = {B_NEG, 0, 0}, brow = {B_NEG, 0, 0}, bcol = {B_NEG, 0, 0};
    int corner = B_NEG;

    // symbols of the lane's cell, one step ahead: (i, j) of step s for this lane
    auto cell_of = [&](int s, int &i, int &j, int &u) {
        const int p = (s + band - d0) & 1;
        u = 2 * x + p;
        const int dl = u - band + d0;                           // j - i
        i = (s - dl) >> 1; j = i + dl;
    };
    auto score_of = [&](int i, int j) -> int {
        if (i < 0 || i >= ql || j < 0 || j >= rl) return 0;
        return mat[map[q[i]] * msize + map[r[j]]];
    };
    int ni, nj, nu;
    cell_of(s_first, ni, nj, nu);
    int sc_next = score_of(ni, nj);

    for (int t = 0; t < nsteps; ++t) {
        const int s = s_first + t;
        int i, j, u;
        cell_of(s, i, j, u);
        const int sc = sc_next;
        cell_of(s + 1, ni, nj, nu);
        sc_next = score_of(ni, nj);                             // independent of this step's arithmetic: its latency is hidden
        const int p = u & 1;
        const bool active = have && s <= s_last && u <= 2 * band && i >= 0 && i < ql && j >= 0 && j < rl;
        const int belowH = b_from_below<LP>(Hm1), belowE = b_from_below<LP>(Em1);
        const int aboveH = b_from_above<LP>(Hm1), aboveF = b_from_above<LP>(Fm1);
        int upH = p ? aboveH : Hm1, upF = p ? aboveF : Fm1;
        int leftH = p ? Hm1 : belowH, leftE = p ? Em1 : belowE;
        int dg = Hm2;
        if (u + 1 > 2 * band || (p && x == LP - 1)) { upH = B_NEG; upF = B_NEG; }          // the cell above lies outside the band
        if (u == 0) { leftH = B_NEG; leftE = B_NEG; }                                      // the cell to the left lies outside the band
        if (i == 0) { upH = rowB(j); upF = B_NEG; dg = j == 0 ? 0 : rowB(j - 1); }
        if (j == 0) { leftH = colB(i); leftE = B_NEG; dg = i == 0 ? 0 : colB(i - 1); }
        int E = max(leftE - ext, leftH - open); if (E < B_NEG) E = B_NEG;
        int F = max(upF - ext, upH - open); if (F < B_NEG) F = B_NEG;
        int H = max(dg + sc, max(E, F));
        if (sw && H < 0) H = 0;
        if (!active) { H = B_NEG; E = B_NEG; F = B_NEG; }
        else {
            const BCand c = {H, i, j};
            if (sw) { if (b_better_sw(c, best)) best = c; }
            else {
                if (i == ql - 1 && j == rl - 1) corner = H;
                if (i == ql - 1 && s2_end && (H > brow.H || (H == brow.H && j < brow.j))) brow = c;
                if (j == rl - 1 && s1_end && (H > bcol.H || (H == bcol.H && i < bcol.i))) bcol = c;
            }
        }
        Hm2 = Hm1; Hm1 = H; Em1 = E; Fm1 = F;
    }

    // ---- reduction over the group ----
#pragma unroll
    for (int off = LP / 2; off >= 1; off >>= 1) {
        BCand o;
        o.H = __shfl_xor(best.H, off, 64); o.i = __shfl_xor(best.i, off, 64); o.j = __shfl_xor(best.j, off, 64);
        if (b_better_sw(o, best)) best = o;
        o.H = __shfl_xor(brow.H, off, 64); o.i = __shfl_xor(brow.i, off, 64); o.j = __shfl_xor(brow.j, off, 64);
        if (o.H > brow.H || (o.H == brow.H && o.j < brow.j)) brow = o;
        o.H = __shfl_xor(bcol.H, off, 64); o.i = __shfl_xor(bcol.i, off, 64); o.j = __shfl_xor(bcol.j, off, 64);
        if (o.H > bcol.H || (o.H == bcol.H && o.i < bcol.i)) bcol = o;
        corner = max(corner, __shfl_xor(corner, off, 64));
    }
    if (x == 0 && have) {
        pmx_record_t rec; rec.flags = 0;
        if (sw) {
            if (best.H == B_NEG) { rec.score = B_NEG; rec.end_query = 0; rec.end_ref = 0; }      // (the band misses the matrix)
            else { rec.score = best.H; rec.end_query = best.i; rec.end_ref = best.j; }
        } else if (mode == PMX_MODE_NW || (!s1_end && !s2_end)) {
            rec.score = corner; rec.end_query = ql - 1; rec.end_ref = rl - 1;         // (-inf when the corner lies outside the band)
        } else {
            BCand res = brow;                                   // B_NEG when the reference end is not free
            if (s1_end && bcol.H > res.H) res = bcol;           // the last column must be strictly better
            rec.score = res.H; rec.end_query = res.i; rec.end_ref = res.j;
        }
        out[pair] = rec;
    }
    }
}

// pairs the band-strip kernel (pmx_bstrip.hip) handed back: a fixed grid over the device-side list
void pmx_banded_retry(int mode, int sg_flags, int open, int ext, const PmxDevMatrix &m, long long n,
                      const uint8_t *qbuf, const int64_t *qoff, int q_shared, const uint8_t *rbuf, const int64_t *roff,
                      int band, const int32_t *diag, const unsigned *list, const int *count, pmx_record_t *out, hipStream_t stream)
{
    const unsigned blocks = (unsigned)std::min<long long>(n, 2048);
#define LBR(LP) hipLaunchKernelGGL((pmx_banded_kernel<LP>), dim3(blocks), dim3(64), 0, stream, \
                                   qbuf, qoff, q_shared, rbuf, roff, n, m.scores, m.mapper, m.msize, mode, sg_flags, open, ext, band, diag, list, count, out)
    if (band <= 15) LBR(16); else if (band <= 31) LBR(32); else LBR(64);
#undef LBR
}

// ---------------------------------------------------------------------------------------------------------------------
// Second form (round 2): the same mapping with both sequences of every pair staged in LDS and a lean interior loop.
//
// pmx_banded_kernel above decides everything per cell (where the cell lies, which neighbour comes from which lane, whether a
// boundary value applies, two global loads for its symbols): ~140 VALU instructions per cell.  Almost all cells of a band lie
// in the interior of the matrix, where none of those decisions is open.  Here
//   * a pair's step counter starts at an even offset of its band, so the parity p(s) is the same in every lane of the wave
//     and the loop is unrolled by it: on even steps the cell to the left comes from lane x - 1 and the cell above is the lane's
//     own previous cell, on odd steps the other way round -- two DPP moves per step, no selects;
//   * both sequences are staged in LDS already mapped (query: symbol * msize as 16 bits, reference: symbol as a byte), so a
//     cell's score is two LDS reads at immediate offsets, one add and one read of the matrix;
//   * the steps are grouped in blocks of eight.  Before the sweep every lane computes the range of blocks in which all its
//     cells are interior cells (rows 1 .. qlen - 2, columns 1 .. rlen - 2); the wave takes the intersection and runs those
//     blocks in the lean loop (H, E and F kept pre-subtracted: E = max(E' , H'), ~16 instructions per cell), everything
//     before and after in the checked step (the logic of the kernel above, reading symbols from LDS);
//   * cells outside the band are never masked in the lean loop: lanes beyond the band compute on "minus infinity" inputs that
//     stay far below every real value, only the band's last lane is forced on odd steps (its odd diagonal is outside).
// Same results as the kernel above, cell for cell (tests/test_gpu_banded.py runs both against the banded oracle).
template <int LP, bool SW>
__global__ __launch_bounds__(64)
void pmx_banded_staged_kernel(const uint8_t *__restrict__ qbuf, const int64_t *__restrict__ qoff, int q_shared,
                              const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff, long long n,
                              const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap, int msize,
                              int mode, int sg_flags, int open, int ext, int band, const int32_t *__restrict__ diag,
                              int QC, int RC /* staging capacity per pair: symbols */, pmx_record_t *__restrict__ out)
{
    __shared__ int16_t mat[PMX_MAX_FAST_MSIZE * PMX_MAX_FAST_MSIZE];
    __shared__ unsigned char map[256];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    for (int x = threadIdx.x; x < msize * msize; x += 64) mat[x] = gmat[x];
    for (int x = threadIdx.x; x < 256; x += 64) map[x] = gmap[x];
    __syncthreads();

    constexpr int NPW = 64 / LP;
    const int lane = threadIdx.x, x = lane % LP, grp = lane / LP;
    const long long pair = (long long)blockIdx.x * NPW + grp;
    const bool have = pair < n;
    const long long pp = have ? pair : n - 1;
    const long long qb = q_shared ? 0 : qoff[pp], rb = roff[pp];
    const int ql = min(q_shared ? q_shared : (int)(qoff[pp + 1] - qb), QC), rl = min((int)(roff[pp + 1] - rb), RC);
    const int d0 = diag ? diag[pp] : 0;
    const bool sg = mode == PMX_MODE_SG;
    const bool s1_end = sg && (sg_flags & PMX_SG_QE), s2_end = sg && (sg_flags & PMX_SG_DE);
    const bool col_pen = mode == PMX_MODE_NW || (sg && !(sg_flags & PMX_SG_QB));
    const bool row_pen = mode == PMX_MODE_NW || (sg && !(sg_flags & PMX_SG_DB));
    auto rowB = [&](int j) -> int { return SW ? 0 : (row_pen ? -(open + j * ext) : 0); };
    auto colB = [&](int i) -> int { return SW ? 0 : (col_pen ? -(open + i * ext) : 0); };

    // ---- stage the mapped sequences: pair g of the wave owns qm[g] (16-bit: symbol * msize) and rm[g] (bytes) --------------
    unsigned short *qm_all = reinterpret_cast<unsigned short *>(dyn);
    unsigned char *rm_all = dyn + (size_t)NPW * QC * 2;
#pragma unroll
    for (int g2 = 0; g2 < NPW; ++g2) {
        const long long qb2 = __shfl(qb, g2 * LP, 64), rb2 = __shfl(rb, g2 * LP, 64);
        const int ql2 = __shfl(ql, g2 * LP, 64), rl2 = __shfl(rl, g2 * LP, 64);
        for (int t = lane; t < ql2; t += 64) qm_all[g2 * QC + t] = (unsigned short)(map[qbuf[qb2 + t]] * msize);
        for (int t = lane; t < rl2; t += 64) rm_all[g2 * RC + t] = map[rbuf[rb2 + t]];
    }
    __syncthreads();
    const unsigned short *qm = qm_all + grp * QC;
    const unsigned char *rm = rm_all + grp * RC;

    // ---- the pair's steps: s = s0 + tau with (s0 + band - d0) even, so the parity of tau is the parity of the diagonal -------
    const int dlo = d0 - band, dhi = d0 + band;
    int s_first = 0;
    if (dlo > 0) s_first = dlo; else if (dhi < 0) s_first = -dhi;
    int s_last = -1;
    {
        int i1 = ql - 1, j1 = rl - 1;
        if (j1 - i1 > dhi) j1 = i1 + dhi; else if (j1 - i1 < dlo) i1 = j1 - dlo;
        if (i1 >= 0 && j1 >= 0 && have) s_last = i1 + j1;
    }
    if (dlo > rl - 1 || dhi < -(ql - 1)) s_last = -1;
    const int s0 = s_first - ((s_first + band - d0) & 1);
    int nsteps = s_last - s0 + 1; if (nsteps < 0) nsteps = 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) nsteps = max(nsteps, __shfl_xor(nsteps, off, 64));
    // lane x at local step tau = 2 m + p: row I0 + m, column J0 + m + p
    const int I0 = ((s0 + band - d0) >> 1) - x, J0 = I0 + 2 * x - band + d0;

    int Hm1 = B_NEG, Em1 = B_NEG, Fm1 = B_NEG, Hm2 = B_NEG;
    BCand best = {B_NEG, 0, 0}, brow = {B_NEG, 0, 0}, bcol = {B_NEG, 0, 0};
    int corner = B_NEG;

    auto checked_step = [&](int tau) {
        const int p = tau & 1, m = tau >> 1;
        const int s = s0 + tau, u = 2 * x + p, i = I0 + m, j = J0 + m + p;
        const bool inside = i >= 0 && i < ql && j >= 0 && j < rl;
        const int sc = inside ? (int)mat[qm[i] + rm[j]] : 0;
        const bool active = have && s <= s_last && u <= 2 * band && inside;
        const int belowH = b_from_below<LP>(Hm1), belowE = b_from_below<LP>(Em1);
        const int aboveH = b_from_above<LP>(Hm1), aboveF = b_from_above<LP>(Fm1);
        int upH = p ? aboveH : Hm1, upF = p ? aboveF : Fm1;
        int leftH = p ? Hm1 : belowH, leftE = p ? Em1 : belowE;
        int dg = Hm2;
        if (u + 1 > 2 * band || (p && x == LP - 1)) { upH = B_NEG; upF = B_NEG; }
        if (u == 0) { leftH = B_NEG; leftE = B_NEG; }
        if (i == 0) { upH = rowB(j); upF = B_NEG; dg = j == 0 ? 0 : rowB(j - 1); }
        if (j == 0) { leftH = colB(i); leftE = B_NEG; dg = i == 0 ? 0 : colB(i - 1); }
        int E = max(leftE - ext, leftH - open); if (E < B_NEG) E = B_NEG;
        int F = max(upF - ext, upH - open); if (F < B_NEG) F = B_NEG;
        int H = max(dg + sc, max(E, F));
        if (SW && H < 0) H = 0;
        if (!active) { H = B_NEG; E = B_NEG; F = B_NEG; }
        else {
            const BCand c = {H, i, j};
            if (SW) { if (b_better_sw(c, best)) best = c; }
            else {
                if (i == ql - 1 && j == rl - 1) corner = H;
                if (i == ql - 1 && s2_end && (H > brow.H || (H == brow.H && j < brow.j))) brow = c;
                if (j == rl - 1 && s1_end && (H > bcol.H || (H == bcol.H && i < bcol.i))) bcol = c;
            }
        }
        Hm2 = Hm1; Hm1 = H; Em1 = E; Fm1 = F;
    };

    // ---- the interior blocks (eight steps each) of this lane, then of the wave --------------------------------------------
    // block B: rows I0 + 4 B .. + 3, columns J0 + 4 B .. + 4, all of them in 1 .. len - 2
    int Blo = -(1 << 28), Bhi = 1 << 28;
    if (x <= band) {
        Blo = max((1 - I0 + 3) >> 2, (1 - J0 + 3) >> 2);
        Bhi = min((ql - 5 - I0) >> 2, (rl - 6 - J0) >> 2);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { Blo = max(Blo, __shfl_xor(Blo, off, 64)); Bhi = min(Bhi, __shfl_xor(Bhi, off, 64)); }
    Blo = max(Blo, 0); Bhi = min(Bhi, (nsteps >> 3) - 1);
    if (band < 1 || Bhi - Blo < 1) { Blo = 0; Bhi = -1; }            // too short for the lean loop to pay

    int tau = 0;
    for (; tau < (Bhi >= Blo ? 8 * Blo : nsteps); ++tau) checked_step(tau);

    if (Bhi >= Blo) {
        // Values are kept biased by 2^30 here ("minus infinity" = B_NEG = -2^30 becomes 0), so a DPP move that zero-fills the
        // lanes without a source delivers "outside the band" by itself: no select at the band's first and last lane.
        constexpr int BIAS = 1 << 30;
        const int keep_odd = x >= band ? 0 : -1;                      // the lane's odd diagonal lies outside the band: forced to 0
        const int floorv = x <= band ? BIAS : 0;                      // (SW) zero floor only inside the band
        const int first_ok = x == 0 ? 0 : -1, last_ok = x == LP - 1 ? 0 : -1;      // LP == 32: the wave-wide shift crosses the groups
        int Hr1 = Hm1 + BIAS, Hr2 = Hm2 + BIAS, Ho1 = Hm1 + BIAS - open, Ee1 = Em1 + BIAS - ext, Fe1 = Fm1 + BIAS - ext;
        int bH = 0, bT = 0;
        const unsigned short *qp = qm + I0 + 4 * Blo;                 // the block's 4 rows
        const unsigned char *rp = rm + J0 + 4 * Blo;                  // and 5 columns (the fifth is the next block's first)
        auto below = [&](int v) -> int {                              // value of lane x - 1, 0 at the band's first lane
            int r = __builtin_amdgcn_update_dpp(0, v, LP == 16 ? 0x111 /*row_shr:1*/ : 0x138 /*wave_shr:1*/, 0xF, 0xF, true);
            if (LP == 32) r &= first_ok;
            return r;
        };
        auto above = [&](int v) -> int {                              // value of lane x + 1, 0 at the group's last lane
            int r = __builtin_amdgcn_update_dpp(0, v, LP == 16 ? 0x101 /*row_shl:1*/ : 0x130 /*wave_shl:1*/, 0xF, 0xF, true);
            if (LP == 32) r &= last_ok;
            return r;
        };
        for (int B = Blo; B <= Bhi; ++B) {
            int sc[8];
            {
                int mq[4], mr[5];
#pragma unroll
                for (int k = 0; k < 4; ++k) mq[k] = qp[k];
#pragma unroll
                for (int k = 0; k < 5; ++k) mr[k] = rp[k];
#pragma unroll
                for (int k = 0; k < 4; ++k) { sc[2 * k] = mat[mq[k] + mr[k]]; sc[2 * k + 1] = mat[mq[k] + mr[k + 1]]; }
            }
            qp += 4; rp += 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                {   // even step: left from lane x - 1, up = own previous cell
                    const int lHo = below(Ho1), lEe = below(Ee1);
                    const int E = max(lEe, lHo), F = max(Fe1, Ho1);
                    int H = max(Hr2 + sc[2 * k], max(E, F));
                    if (SW) { H = max(H, floorv); if (H > bH) { bH = H; bT = 8 * B + 2 * k; } }
                    Hr2 = Hr1; Hr1 = H; Ho1 = H - open; Ee1 = E - ext; Fe1 = F - ext;
                }
                {   // odd step: up from lane x + 1, left = own previous cell
                    const int uHo = above(Ho1), uFe = above(Fe1);
                    int E = max(Ee1, Ho1), F = max(uFe, uHo);
                    int H = max(Hr2 + sc[2 * k + 1], max(E, F));
                    if (SW) H = max(H, floorv);
                    H &= keep_odd; E &= keep_odd; F &= keep_odd;      // (nothing real may leak to the lanes beyond the band)
                    if (SW) { if (H > bH) { bH = H; bT = 8 * B + 2 * k + 1; } }
                    Hr2 = Hr1; Hr1 = H; Ho1 = H - open; Ee1 = E - ext; Fe1 = F - ext;
                }
            }
        }
        // back to the plain domain (anything at or below "minus infinity" is minus infinity)
        auto plain = [&](int v) -> int { return v <= 0 ? B_NEG : v - BIAS; };
        Hm1 = plain(Hr1); Hm2 = plain(Hr2); Em1 = plain(Ee1 + ext); Fm1 = plain(Fe1 + ext);
        if (SW && have && bH - BIAS > best.H) {                       // (interior cells come after every earlier candidate of the lane)
            best.H = bH - BIAS; best.i = I0 + (bT >> 1); best.j = J0 + (bT >> 1) + (bT & 1);
        }
        tau = 8 * (Bhi + 1);
    }
    for (; tau < nsteps; ++tau) checked_step(tau);

    // ---- reduction over the group (as in the kernel above) -------------------------------------------------------------------
#pragma unroll
    for (int off = LP / 2; off >= 1; off >>= 1) {
        BCand o;
        o.H = __shfl_xor(best.H, off, 64); o.i = __shfl_xor(best.i, off, 64); o.j = __shfl_xor(best.j, off, 64);
        if (b_better_sw(o, best)) best = o;
        o.H = __shfl_xor(brow.H, off, 64); o.i = __shfl_xor(brow.i, off, 64); o.j = __shfl_xor(brow.j, off, 64);
        if (o.H > brow.H || (o.H == brow.H && o.j < brow.j)) brow = o;
        o.H = __shfl_xor(bcol.H, off, 64); o.i = __shfl_xor(bcol.i, off, 64); o.j = __shfl_xor(bcol.j, off, 64);
        if (o.H > bcol.H || (o.H == bcol.H && o.i < bcol.i)) bcol = o;
        corner = max(corner, __shfl_xor(corner, off, 64));
    }
    if (x == 0 && have) {
        pmx_record_t rec; rec.flags = 0;
        if (SW) {
            if (best.H == B_NEG) { rec.score = B_NEG; rec.end_query = 0; rec.end_ref = 0; }
            else { rec.score = best.H; rec.end_query = best.i; rec.end_ref = best.j; }
        } else if (mode == PMX_MODE_NW || (!s1_end && !s2_end)) {
            rec.score = corner; rec.end_query = ql - 1; rec.end_ref = rl - 1;
        } else {
            BCand res = brow;
            if (s1_end && bcol.H > res.H) res = bcol;
            rec.score = res.H; rec.end_query = res.i; rec.end_ref = res.j;
        }
        out[pair] = rec;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Third form (round 3): LOCAL alignment, two pairs per lane group in the int16 halves of every register, the lean loop only.
//
// The staged kernel above spends 17 instructions per cell in its interior loop and ~140 in the checked steps at the matrix edges.
// For local alignment the edges need no checking at all: a cell outside the matrix that computes on PAD symbols (score -open
// against everything) floors to 0 like the boundary it stands for, feeds nothing real (dependencies run down and right) and can
// never strictly exceed a real cell (its value reaches it through a gap) -- so both sequences are staged with pad margins on
// both sides and EVERY step runs the lean form.  Values are biased (true 0 = B, B = 1024 + open + extend: every live value is
// 0 or in [1024, 31743], where v_pk_maximum3_f16 is an exact integer max3 and v_pk_sub_u16 with clamp keeps "minus infinity" = 0
// sticky); the strips carry H - open and the scores carry + open, so the diagonal sum is one add; the zero floor is folded into
// E.  A lane keeps its best and the step where it was first strictly exceeded (per half); cell (i, j) follows from the step.
// 15.5 instructions per TWO cells.  The host proves the range (score bound + B < 31744, min score + open >= 0).
typedef short b_v2s __attribute__((ext_vector_type(2)));
typedef unsigned short b_v2us __attribute__((ext_vector_type(2)));
typedef _Float16 b_v2h __attribute__((ext_vector_type(2)));
#define BPK(x) __builtin_bit_cast(b_v2s, (int)(x))
#define BI32(x) __builtin_bit_cast(int, (x))
__device__ __forceinline__ int bp_max3(int a, int b, int c)
{
    const b_v2h r = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(b_v2h, a), __builtin_bit_cast(b_v2h, b)),
                                                  __builtin_bit_cast(b_v2h, c));
    return __builtin_bit_cast(int, r);
}
__device__ __forceinline__ int bp_subus(int a, int b)         // v_pk_sub_u16 clamp: saturates at 0
{
    return __builtin_bit_cast(int, __builtin_elementwise_sub_sat(__builtin_bit_cast(b_v2us, a), __builtin_bit_cast(b_v2us, b)));
}

// FORM 0: a byte lookup in the padded matrix per cell (any alphabet up to PMX_MAX_FAST_MSIZE - 1 letters).
// FORM 1 (<= 7 letters + the pad symbol): a matrix row is 8 bytes -- one ds_read_b64 per query symbol and a v_perm_b32 by the
//   reference symbol replace the byte lookup (fewer address adds; measured 40.0 -> 38.3 ms on cfg 5's second pass).
// FORM 2 (<= 7 letters, ONE shared query): the two pairs of a lane group are started on the SAME query row, so a lane's row of the
//   matrix is one register pair for both of them and ONE v_perm_b32 -- selector {symbol A | 0x0C00, symbol B | 0x0C00} -- yields both
//   halves' scores.  The query is staged as its matrix rows (8 bytes a position), once per workgroup of four waves; the reference
//   windows as 16-bit selectors.  The pair whose band enters the matrix further down starts early on cells outside the matrix
//   (pad symbols: zeros that feed nothing); the processing order (pmx_sort.hip) puts pairs with nearby entry rows side by side.
template <int LP, int FORM>
__global__ __launch_bounds__(FORM == 2 ? 256 : 64)
void pmx_banded_packed_kernel(const uint8_t *__restrict__ qbuf, const int64_t *__restrict__ qoff, int q_shared,
                              const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff, long long n,
                              const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap, int msize,
                              int open, int ext, int band, const int32_t *__restrict__ diag,
                              int QC, int RC /* staging capacity per pair, margins included */,
                              const unsigned *__restrict__ perm /* optional processing order: position -> pair */, pmx_record_t *__restrict__ out)
{
    constexpr bool SMALL = FORM == 1, QS = FORM == 2;
    constexpr int NW = QS ? 4 : 1, NT = 64 * NW;             // waves, threads per workgroup
    constexpr int MG = 160;                                  // pad symbols in front of and behind every staged sequence
    __shared__ unsigned char matp[(PMX_MAX_FAST_MSIZE + 1) * (PMX_MAX_FAST_MSIZE + 1)];    // score + open as a byte; row / column msize = the pad symbol
    __shared__ unsigned char map[256];
    __shared__ __attribute__((aligned(8))) unsigned char mrow[8 * 8];             // SMALL: row a = the 8 score bytes of query symbol a
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int MS1 = msize + 1;
    const int QMUL = SMALL ? 8 : MS1;                        // what a staged query symbol is multiplied by (the row's byte offset)
    for (int x = threadIdx.x; x < MS1 * MS1; x += NT) {
        const int a = x / MS1, b = x - a * MS1;
        matp[x] = (a < msize && b < msize) ? (unsigned char)(gmat[a * msize + b] + open) : (unsigned char)0;      // pad: score -open
    }
    if ((SMALL || QS) && threadIdx.x < 64) {
        const int a = threadIdx.x >> 3, b = threadIdx.x & 7;
        mrow[threadIdx.x] = (a < msize && b < msize) ? (unsigned char)(gmat[a * msize + b] + open) : (unsigned char)0;
    }
    for (int x = threadIdx.x; x < 256; x += NT) map[x] = gmap[x];
    __syncthreads();

    constexpr int NG = 64 / LP, NPW = 2 * NG;                // lane groups, pairs per wave
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, x = lane % LP, grp = lane / LP;
    // per half: geometry of the pair
    long long pairH[2], qbH[2], rbH[2]; bool haveH[2]; int qlH[2], rlH[2], d0H[2], s0H[2], I0H[2], J0H[2], nstH[2], cwH[2], wlenH[2]; bool hitH[2];
    int nsteps = 0;
    // LDS: FORM 0/1 [NPW][QC] 16-bit query symbols (x row stride), [NPW][RC] reference bytes;
    //      FORM 2   [QC] 8-byte matrix rows of the one query, [NW][NPW][RC] 16-bit reference selectors
    unsigned short *qm_all = reinterpret_cast<unsigned short *>(dyn);
    unsigned char *rm_all = dyn + (size_t)NPW * QC * 2;
    uint2 *qrows = reinterpret_cast<uint2 *>(dyn);
    unsigned short *rs_all = reinterpret_cast<unsigned short *>(dyn + (size_t)QC * 8) + (size_t)wave * NPW * RC;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long long pos = ((long long)blockIdx.x * NW + wave) * NPW + 2 * grp + h;
        haveH[h] = pos < n;
        const long long pp = perm ? (long long)perm[haveH[h] ? pos : n - 1] : (haveH[h] ? pos : n - 1);
        pairH[h] = pp;
        const long long qb = q_shared ? 0 : qoff[pp], rb = roff[pp];
        const int ql = min(q_shared ? q_shared : (int)(qoff[pp + 1] - qb), QC - 2 * MG), rl = (int)(roff[pp + 1] - rb);   // (the reference is staged by window)
        qlH[h] = ql; rlH[h] = rl; qbH[h] = qb; rbH[h] = rb;
        const int d0 = diag ? diag[pp] : 0;
        d0H[h] = d0;
        const int dlo = d0 - band, dhi = d0 + band;
        int s_first = 0;
        if (dlo > 0) s_first = dlo; else if (dhi < 0) s_first = -dhi;
        int s_last = -1;
        {
            int i1 = ql - 1, j1 = rl - 1;
            if (j1 - i1 > dhi) j1 = i1 + dhi; else if (j1 - i1 < dlo) i1 = j1 - dlo;
            if (i1 >= 0 && j1 >= 0 && haveH[h]) s_last = i1 + j1;
        }
        if (dlo > rl - 1 || dhi < -(ql - 1)) s_last = -1;
        hitH[h] = s_last >= 0;
        // two steps early (same parity): every lane starts on pad cells, whose zeros ARE the boundary the first real cells read
        const int s0 = s_first - ((s_first + band - d0) & 1) - 2;
        int ns = s_last - s0 + 1; if (ns < 0 || s_last < 0) ns = 0;
        s0H[h] = s0; nstH[h] = ns;
    }
    if (QS) {
        // one query row per lane for both halves: the half whose band enters the matrix further down starts that much earlier
        const int a0 = (s0H[0] + band - d0H[0]) >> 1, a1 = (s0H[1] + band - d0H[1]) >> 1;
        const int a = (nstH[0] && nstH[1]) ? min(a0, a1) : nstH[0] ? a0 : a1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ah = h ? a1 : a0;
            if (nstH[h]) { s0H[h] -= 2 * (ah - a); nstH[h] += 2 * (ah - a); }
            else s0H[h] = 2 * a - band + d0H[h];             // (no cell of its own: it rides along on the other half's rows)
        }
    }
    if (QS) {                                                // the query's matrix rows, once per workgroup
        const int ql = qlH[0];
        for (int t = threadIdx.x; t < ql + 2 * MG; t += NT) {
            const int i = t - MG;
            const int sym = (i >= 0 && i < ql) ? (int)map[qbuf[i]] : msize;
            qrows[t] = *reinterpret_cast<const uint2 *>(mrow + 8 * sym);
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int s0 = s0H[h], d0 = d0H[h], ns = nstH[h], ql = qlH[h], rl = rlH[h];
        const long long qb = qbH[h], rb = rbH[h];
        nsteps = max(nsteps, ns);
        I0H[h] = ((s0 + band - d0) >> 1) - x; J0H[h] = I0H[h] + 2 * x - band + d0;
        // Only the WINDOW of the reference the band crosses is staged: columns cw .. cw + band + steps / 2 (lane x starts at column
        // cw + x and moves one column per two steps) -- about qlen + 2 band symbols of a reference that may be five times as long; the
        // LDS this saves is occupancy.  Position t of the staged window = column cw - 8 + t.
        const int cw = ((s0 + band - d0) >> 1) - band + d0;
        cwH[h] = cw;
        const int wlen = min(band + (ns >> 1) + 24, RC - 16);
        // stage this pair's sequences (all lanes of the wave work on one pair at a time: its offsets come from its group's lane 0)
#pragma unroll
        for (int g2 = 0; g2 < NG; ++g2) {
            const long long qb2 = __shfl(qb, g2 * LP, 64), rb2 = __shfl(rb, g2 * LP, 64);
            const int ql2 = __shfl(ql, g2 * LP, 64), rl2 = __shfl(rl, g2 * LP, 64);
            const int cw2 = __shfl(cw, g2 * LP, 64), wl2 = __shfl(wlen, g2 * LP, 64);
            if (!QS) {
                unsigned short *qd = qm_all + (size_t)(2 * g2 + h) * QC;
                if (!(q_shared && (g2 || h))) {              // one shared query: staged once, every pair reads that copy
                    for (int t = lane; t < ql2 + 2 * MG; t += 64) {
                        const int i = t - MG;
                        qd[t] = (unsigned short)(((i >= 0 && i < ql2) ? (int)map[qbuf[qb2 + i]] : msize) * QMUL);
                    }
                }
                unsigned char *rd = rm_all + (size_t)(2 * g2 + h) * RC;
                for (int t = lane; t < wl2 + 16; t += 64) {
                    const int j = cw2 - 8 + t;
                    rd[t] = (unsigned char)((j >= 0 && j < rl2 && t < wl2 + 8) ? (int)map[rbuf[rb2 + j]] : msize);
                }
            } else {
                unsigned short *rd = rs_all + (size_t)(2 * g2 + h) * RC;
                for (int t = lane; t < wl2 + 16; t += 64) {
                    const int j = cw2 - 8 + t;
                    rd[t] = (unsigned short)(0x0C00 | ((j >= 0 && j < rl2 && t < wl2 + 8) ? (int)map[rbuf[rb2 + j]] : msize));
                }
            }
        }
        wlenH[h] = wlen;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) nsteps = max(nsteps, __shfl_xor(nsteps, off, 64));
    nsteps = (nsteps + 7) & ~7;
    __syncthreads();

    const int B = 1024 + open + ext;                          // true 0
    const int B2 = B * 0x00010001, vOpen = open * 0x00010001, vExt = ext * 0x00010001;
    int bH = B2, bT = 0;                                      // best (only a score above 0 is tracked) and the step where it was first exceeded
    const int keep_odd = x >= band ? 0 : -1;                  // the lane's odd diagonal lies outside the band: forced to "minus infinity"
    const int floorv = x <= band ? B2 : 0;                    // zero floor only inside the band
    const int first_ok = x == 0 ? 0 : -1, last_ok = x == LP - 1 ? 0 : -1;
    auto below = [&](int v) -> int {                          // value of lane x - 1, 0 ("minus infinity") at the band's first lane
        int r = __builtin_amdgcn_mov_dpp(v, LP == 16 ? 0x111 /*row_shr:1*/ : 0x138 /*wave_shr:1*/, 0xF, 0xF, true);      // (bound_ctrl: no `old` register)
        if (LP == 32) r &= first_ok;
        return r;
    };
    auto above = [&](int v) -> int {                          // value of lane x + 1, 0 at the group's last lane
        int r = __builtin_amdgcn_mov_dpp(v, LP == 16 ? 0x101 /*row_shl:1*/ : 0x130 /*wave_shl:1*/, 0xF, 0xF, true);
        if (LP == 32) r &= last_ok;
        return r;
    };
    const b_v2s sh15 = {15, 15};
    int signs = (int)0x80008000;
    asm volatile("" : "+v"(signs));
    // best and its step, per half, once per PAIR of steps (a lane's even cell and the odd cell after it): 0xFFFF where either exceeds
    // the best (one max3, one packed subtract, one shift); what is inserted under that mask is the pair's number with bit 15 =
    // "the odd cell is the larger" (sign of He - Ho: on a tie the even cell, the earlier one, stands).  6 instructions per two
    // steps; one track per step was 8.
    auto track2 = [&](int He, int Ho, int tau2) {
        const int P = bp_max3(bH, He, Ho);
        const int m = BI32((BPK(bH) - BPK(P)) >> sh15);
        const int d = BI32(BPK(He) - BPK(Ho));
        int tw;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(tw) : "v"(d), "v"(signs), "s"(tau2 * 0x00010001));      // (one SGPR per instruction)
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(bT) : "v"(m), "v"(tw), "v"(bT));
        bH = P;
    };
    // LDS read positions (element indices), clamped into the trailing pad once a pair has run past its sequences
    int qpos[2], rpos[2], qend[2], rend[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        qpos[h] = I0H[h] + MG; rpos[h] = J0H[h] - cwH[h] + 8;                     // (reference: position in the staged window)
        qend[h] = qlH[h] + MG + 8; rend[h] = wlenH[h] + 8;                        // 8 or more pads follow
        qpos[h] = min(max(qpos[h], 0), qend[h]); rpos[h] = min(max(rpos[h], 0), rend[h]);
        // lanes beyond the band read pad symbols only: on real symbols their unfloored, unmasked cells could climb along a run of
        // matches (low-complexity sequences) from "minus infinity" into the range of real values -- B is 1 024, not 2^30
        if (x > band) { qpos[h] = qend[h]; rpos[h] = rend[h]; }
    }
    const unsigned short *qmA = q_shared ? qm_all : qm_all + (size_t)(2 * grp) * QC, *qmB = q_shared ? qm_all : qmA + QC;
    const unsigned char *rmA = rm_all + (size_t)(2 * grp) * RC, *rmB = rmA + RC;
    const unsigned short *rsA = rs_all + (size_t)(2 * grp) * RC, *rsB = rsA + RC;

    // state: everything "minus infinity" (0) except nothing -- the pad cells in front of the matrix produce the zero boundary
    int Ho1 = 0, Ho2 = 0, Ee1 = 0, Fe1 = 0;                   // H - open of the lane's previous cell and of the one before; E - ext; F - ext
    // (fetching a block's scores one block ahead was measured and lost: 53.3 -> 58.9 ms on cfg 5's second pass -- the other waves
    //  of the SIMD already cover the two dependent LDS reads, the extra register moves cost more)
    // (the kernel is LDS-bound: a block's fifth reference symbol is the next block's first and is carried over, not read again;
    //  once a position is clamped into the trailing pads the carried symbol is a pad as well)
    int rcarry[2] = {QS ? (int)rsA[rpos[0]] : (int)rmA[rpos[0]], QS ? (int)rsB[rpos[1]] : (int)rmB[rpos[1]]};
    auto fetch = [&](int (&dst)[8]) {
        int mq[2][4], mr[2][5];
        mr[0][0] = rcarry[0]; mr[1][0] = rcarry[1];
#pragma unroll
        for (int k = 1; k < 5; ++k) { mr[0][k] = QS ? (int)rsA[rpos[0] + k] : (int)rmA[rpos[0] + k]; mr[1][k] = QS ? (int)rsB[rpos[1] + k] : (int)rmB[rpos[1] + k]; }
        rcarry[0] = mr[0][4]; rcarry[1] = mr[1][4];
        if (QS) {
            unsigned sel[5];                                 // {symbol A | 0x0C00, symbol B | 0x0C00}: bytes 1 and 3 of a score stay 0
#pragma unroll
            for (int c = 0; c < 5; ++c) sel[c] = ((unsigned)mr[1][c] << 16) | (unsigned)mr[0][c];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint2 row = qrows[qpos[0] + k];
                dst[2 * k] = (int)__builtin_amdgcn_perm(row.y, row.x, sel[k]);
                dst[2 * k + 1] = (int)__builtin_amdgcn_perm(row.y, row.x, sel[k + 1]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) rpos[h] = min(rpos[h] + 4, rend[h]);
            qpos[0] = min(qpos[0] + 4, qend[0]);
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { mq[0][k] = qmA[qpos[0] + k]; mq[1][k] = qmB[qpos[1] + k]; }
        if (SMALL) {
            unsigned selA[5], selB[5];                       // the reference symbol picks its byte of the row: pair A into byte 0, pair B into byte 2
#pragma unroll
            for (int c = 0; c < 5; ++c) { selA[c] = (unsigned)mr[0][c] | 0x0C0C0C00u; selB[c] = ((unsigned)mr[1][c] << 16) | 0x0C000C0Cu; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint2 ra = *reinterpret_cast<const uint2 *>(mrow + mq[0][k]), rb2 = *reinterpret_cast<const uint2 *>(mrow + mq[1][k]);
                dst[2 * k] = (int)(__builtin_amdgcn_perm(ra.y, ra.x, selA[k]) | __builtin_amdgcn_perm(rb2.y, rb2.x, selB[k]));
                dst[2 * k + 1] = (int)(__builtin_amdgcn_perm(ra.y, ra.x, selA[k + 1]) | __builtin_amdgcn_perm(rb2.y, rb2.x, selB[k + 1]));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dst[2 * k] = (int)matp[mq[0][k] + mr[0][k]] | ((int)matp[mq[1][k] + mr[1][k]] << 16);
                dst[2 * k + 1] = (int)matp[mq[0][k] + mr[0][k + 1]] | ((int)matp[mq[1][k] + mr[1][k + 1]] << 16);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) { qpos[h] = min(qpos[h] + 4, qend[h]); rpos[h] = min(rpos[h] + 4, rend[h]); }
    };
    int sc[8];
    for (int t0 = 0; t0 < nsteps; t0 += 8) {
        fetch(sc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int He;
            {   // even step: left from lane x - 1, up = own previous cell
                const int lHo = below(Ho1), lEe = below(Ee1);
                const int E = bp_max3(lEe, lHo, floorv), F = bp_max3(Fe1, Ho1, Ho1);
                const int H = bp_max3(Ho2 + sc[2 * k], E, F);
                He = H;
                Ho2 = Ho1; Ho1 = bp_subus(H, vOpen); Ee1 = bp_subus(E, vExt); Fe1 = bp_subus(F, vExt);
            }
            {   // odd step: up from lane x + 1, left = own previous cell
                const int uHo = above(Ho1), uFe = above(Fe1);
                int E = bp_max3(Ee1, Ho1, floorv);
                const int F = bp_max3(uFe, uHo, uHo);
                int H = bp_max3(Ho2 + sc[2 * k + 1], E, F);
                // the band's last lane: its odd diagonal is outside.  H and E forced to "minus infinity" keep every lane beyond it at
                // 0 for good (they read pad symbols, and nothing else feeds them), so the F that comes back from there is 0 unforced
                H &= keep_odd; E &= keep_odd;
                track2(He, H, (t0 >> 1) + k);
                Ho2 = Ho1; Ho1 = bp_subus(H, vOpen); Ee1 = bp_subus(E, vExt); Fe1 = bp_subus(F, vExt);
            }
        }
    }

    // ---- per half: the lane's candidate, reduced over the group -----------------------------------------------------------
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int hb = h ? (int)((unsigned)bH >> 16) : (bH & 0xFFFF), tb = h ? (int)((unsigned)bT >> 16) : (bT & 0xFFFF);
        BCand best = {B_NEG, 0, 0};
        if (hb > B) { best.H = hb - B; best.i = I0H[h] + (tb & 0x7FFF); best.j = J0H[h] + (tb & 0x7FFF) + (tb >> 15); }
#pragma unroll
        for (int off = LP / 2; off >= 1; off >>= 1) {
            BCand o;
            o.H = __shfl_xor(best.H, off, 64); o.i = __shfl_xor(best.i, off, 64); o.j = __shfl_xor(best.j, off, 64);
            if (b_better_sw(o, best)) best = o;
        }
        if (x == 0 && haveH[h]) {
            pmx_record_t rec; rec.flags = 0;
            if (!hitH[h]) { rec.score = B_NEG; rec.end_query = 0; rec.end_ref = 0; }        // the band misses the matrix
            else if (best.H == B_NEG) {
                // no cell above 0: the first cell of the band in column-major order (smallest column, then smallest row)
                const int dlo = d0H[h] - band, dhi = d0H[h] + band;
                const int j = max(0, dlo), i = max(0, j - dhi);
                rec.score = 0; rec.end_query = i; rec.end_ref = j;
            } else { rec.score = best.H; rec.end_query = best.i; rec.end_ref = best.j; }
            out[pairH[h]] = rec;
        }
    }
}

int pmx_launch_banded(int mode, int sg_flags, int open, int ext, const PmxDevMatrix &m, long long n,
                      const uint8_t *qbuf, const int64_t *qoff, int q_shared, const uint8_t *rbuf, const int64_t *roff,
                      int max_qlen, int max_rlen, int band, const int32_t *diag, pmx_record_t *out, hipStream_t stream,
                      const char **kernel_name, void *sort_scratch, unsigned *retry_list, int *retry_count)
{
    if (n <= 0) return 0;
    if (pmx_env("PMX_NO_FAST_BANDED")) return 1;
    // the band-strip kernel (band coordinates, packed int16, C offsets per lane) where its window holds
    if (!pmx_env("PMX_BANDED_NO_STAGING") && !pmx_env("PMX_BANDED_NO_PACKED")) {
        const int rcs = pmx_launch_bstrip(mode, sg_flags, open, ext, m, n, qbuf, qoff, q_shared, rbuf, roff, max_qlen, max_rlen, band, diag, out, stream,
                                          kernel_name, sort_scratch, retry_list, retry_count);
        if (rcs <= 0) return rcs;
    }
    if (m.msize > PMX_MAX_FAST_MSIZE || band > 63) return 1;       // wider bands: the general kernel masks instead
    const int LPs = band <= 15 ? 16 : band <= 31 ? 32 : 64, NPW = 64 / LPs;
    const int QC = (max_qlen + 3) & ~3, RC = (max_rlen + 3) & ~3;
    const size_t lds = (size_t)NPW * ((size_t)QC * 2 + RC);
    // staged form: both sequences of the wave's pairs fit the LDS, and "minus infinity" cannot drift into the range of real values
    const bool staged = !pmx_env("PMX_BANDED_NO_STAGING") && lds <= 60 * 1024 && open <= 512 && ext <= 512 &&
                        (long long)max_qlen + max_rlen < (1 << 20);
    const bool sw = mode == PMX_MODE_SW;
    // local alignment inside the int16 window: two pairs per lane group, lean loop only (third form above)
    if (sw && !pmx_env("PMX_BANDED_NO_STAGING") && !pmx_env("PMX_BANDED_NO_PACKED") && open >= 0 && ext >= 0 && open + ext <= 2048 &&
        m.min + open >= 0 && m.max + open <= 255 && m.msize <= PMX_MAX_FAST_MSIZE - 1 &&
        (long long)(max_qlen < max_rlen ? max_qlen : max_rlen) + band + 8 < 32000 &&       // a pair of steps is numbered in 15 bits

        (long long)(max_qlen < max_rlen ? max_qlen : max_rlen) * (m.max > 0 ? m.max : 0) + 1024 + open + ext + (m.max > 0 ? m.max : 0) < 31000) {
        const int LPp = band <= 15 ? 16 : band <= 31 ? 32 : 64, NPWp = 2 * (64 / LPp);
        // (reference: only the window the band crosses is staged -- at most band + steps / 2 + 24 symbols, steps <= 2 (qlen + band) + 4)
        const int QCp = ((max_qlen + 2 * 160 + 3) & ~3), RCp = ((std::min(max_rlen, max_qlen + 2 * band + 8) + band + 48 + 3) & ~3);
        const size_t ldsp = (size_t)NPWp * ((size_t)QCp * 2 + RCp);
        if (ldsp <= 150 * 1024) {
            // neighbours in one lane group run for the longer of their two bands: process the pairs in the order of their bands' lengths
            // one shared query over a small alphabet: FORM 2 (both pairs of a lane group on the same query rows).  Its reference windows
            // span the query's rows whatever the reference's length (the pair that starts early walks pad columns), 16 bits a symbol
            const int RCq = (max_qlen + 3 * band + 56 + 3) & ~3;
            const size_t ldsq = (size_t)QCp * 8 + (size_t)4 * NPWp * RCq * 2;
            const bool shared_rows = q_shared && m.msize <= 7 && ldsq <= 40 * 1024 && (long long)max_qlen + band + 8 < 32000 &&
                                     !pmx_env("PMX_BANDED_NO_ROWPERM") && !pmx_env("PMX_BANDED_NO_SHARED_ROWS");
            const unsigned *perm = nullptr;
            if (sort_scratch && n >= 4096) {
                const int rc = pmx_build_band_perm(qoff, q_shared, roff, diag, band, n, sort_scratch, &perm, stream, shared_rows);
                if (rc < 0) return rc;
            }
#define LPK(LP, FORM, NWV, LDSB, RCV) do { if ((LDSB) > 48 * 1024) { const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_banded_packed_kernel<LP, FORM>), 156 * 1024); if (rc) return rc; } \
        hipLaunchKernelGGL((pmx_banded_packed_kernel<LP, FORM>), dim3((unsigned)((n + NPWp * (NWV) - 1) / (NPWp * (NWV)))), dim3(64 * (NWV)), (LDSB), stream, \
                           qbuf, qoff, q_shared, rbuf, roff, n, m.scores, m.mapper, m.msize, open, ext, band, diag, QCp, (RCV), perm, out); } while (0)
            if (shared_rows) { if (band <= 15) LPK(16, 2, 4, ldsq, RCq); else if (band <= 31) LPK(32, 2, 4, ldsq, RCq); else LPK(64, 2, 4, ldsq, RCq); }
            else if (m.msize <= 7 && !pmx_env("PMX_BANDED_NO_ROWPERM")) { if (band <= 15) LPK(16, 1, 1, ldsp, RCp); else if (band <= 31) LPK(32, 1, 1, ldsp, RCp); else LPK(64, 1, 1, ldsp, RCp); }
            else { if (band <= 15) LPK(16, 0, 1, ldsp, RCp); else if (band <= 31) LPK(32, 0, 1, ldsp, RCp); else LPK(64, 0, 1, ldsp, RCp); }
#undef LPK
            if (kernel_name) *kernel_name = shared_rows ? "pmx_banded_packed_kernel/shared query rows"
                                           : (m.msize <= 7 && !pmx_env("PMX_BANDED_NO_ROWPERM")) ? "pmx_banded_packed_kernel/matrix rows" : "pmx_banded_packed_kernel";
            const hipError_t e = hipGetLastError();
            return e == hipSuccess ? 0 : -(int)e;
        }
    }
#define LB(LP) hipLaunchKernelGGL((pmx_banded_kernel<LP>), dim3((unsigned)((n + 64 / LP - 1) / (64 / LP))), dim3(64), 0, stream, \
                                  qbuf, qoff, q_shared, rbuf, roff, n, m.scores, m.mapper, m.msize, mode, sg_flags, open, ext, band, diag, nullptr, nullptr, out)
#define LS(LP, SWF) hipLaunchKernelGGL((pmx_banded_staged_kernel<LP, SWF>), dim3((unsigned)((n + 64 / LP - 1) / (64 / LP))), dim3(64), lds, stream, \
                                  qbuf, qoff, q_shared, rbuf, roff, n, m.scores, m.mapper, m.msize, mode, sg_flags, open, ext, band, diag, QC, RC, out)
    if (staged) {
        if (band <= 15) { if (sw) LS(16, true); else LS(16, false); }
        else if (band <= 31) { if (sw) LS(32, true); else LS(32, false); }
        else { if (sw) LS(64, true); else LS(64, false); }
    } else {
        if (band <= 15) LB(16);
        else if (band <= 31) LB(32);
        else LB(64);
    }
#undef LB
#undef LS
    if (kernel_name) *kernel_name = staged ? "pmx_banded_staged_kernel" : "pmx_banded_kernel";
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
