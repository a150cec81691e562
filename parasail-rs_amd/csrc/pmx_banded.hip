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
                       pmx_record_t *__restrict__ out)
{
    __shared__ int16_t mat[PMX_MAX_FAST_MSIZE * PMX_MAX_FAST_MSIZE];
    __shared__ unsigned char map[256];
    for (int x = threadIdx.x; x < msize * msize; x += 64) mat[x] = gmat[x];
    for (int x = threadIdx.x; x < 256; x += 64) map[x] = gmap[x];
    __syncthreads();

    constexpr int NPW = 64 / LP;
    const int lane = threadIdx.x, x = lane % LP;
    const long long pair = (long long)blockIdx.x * NPW + lane / LP;
    const bool have = pair < n;
    const long long pp = have ? pair : n - 1;
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

int pmx_launch_banded(int mode, int sg_flags, int open, int ext, const PmxDevMatrix &m, long long n,
                      const uint8_t *qbuf, const int64_t *qoff, int q_shared, const uint8_t *rbuf, const int64_t *roff,
                      int max_qlen, int max_rlen, int band, const int32_t *diag, pmx_record_t *out, hipStream_t stream)
{
    (void)max_qlen; (void)max_rlen;
    if (n <= 0) return 0;
    if (pmx_env("PMX_NO_FAST_BANDED")) return 1;
    if (m.msize > PMX_MAX_FAST_MSIZE || band > 63) return 1;       // wider bands: the general kernel masks instead
#define LB(LP) hipLaunchKernelGGL((pmx_banded_kernel<LP>), dim3((unsigned)((n + 64 / LP - 1) / (64 / LP))), dim3(64), 0, stream, \
                                  qbuf, qoff, q_shared, rbuf, roff, n, m.scores, m.mapper, m.msize, mode, sg_flags, open, ext, band, diag, out)
    if (band <= 15) LB(16);
    else if (band <= 31) LB(32);
    else LB(64);
#undef LB
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
