// pmx_table.hip -- score tables and last rows / columns at HBM write speed.  gfx950 only.
//
// The accessors Alignment::get_score_table / get_score_row / get_score_col
// (/root/reference/src/alignment/mod.rs:123-288) hand out int32 [query_len][ref_len] tables, rows of ref_len and columns of
// query_len entries (layout: src/alignment/table.rs:4-9).  Dispatch names `*_table_*` / `*_rowcol_*` without statistics
// (src/aligner/mod.rs:319-329).  A score table is 4 bytes per cell: unlike every other output of the path this one IS bound by
// HBM write bandwidth, so the mapping is chosen for the store, not for the arithmetic:
//
//   * one wave per pair, the table is produced ROW BY ROW; lane l owns the C consecutive columns l*C .. l*C+C-1, so a row
//     leaves as C/4 16-byte stores per lane, 64 contiguous bytes per lane, 4 KB contiguous per wave and row (C = 16);
//   * vertical dependencies (F chain, H of the row above) stay in the lane's registers, the diagonal predecessor of a lane's
//     first column comes from lane l-1 with one DPP move per row;
//   * the horizontal dependency (E chain along the row) is a prefix max-scan with linear decay ("scan" formulation): a local
//     pass gives every lane the E that leaves its columns if nothing came in, a 6-step DPP prefix-max over the 64 lanes (values
//     un-decayed by l*C*ext) gives the E that enters every lane, a second local pass applies it.  Exact for open >= extend
//     (closing and reopening a gap never beats extending it), which the reference asks for (src/aligner/mod.rs:139-153).
//   * 32-bit lanes: no saturation, every mode (nw / sg with any free ends / sw), end positions with the oracle's rules.
#include "pmx_common.h"
#include "pmx_switches.h"
#include <cstdlib>

#define TNEG (INT32_MIN / 2)

__device__ __forceinline__ int t_lane_below(int x, int fill)          // value of lane - 1, `fill` in lane 0
{
    return __builtin_amdgcn_update_dpp(fill, x, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
}
// inclusive prefix maximum over the 64 lanes
__device__ __forceinline__ int t_prefix_max(int v)
{
    int t;
    t = __builtin_amdgcn_update_dpp(TNEG, v, 0x111 /*row_shr:1*/, 0xF, 0xF, false); v = max(v, t);
    t = __builtin_amdgcn_update_dpp(TNEG, v, 0x112 /*row_shr:2*/, 0xF, 0xF, false); v = max(v, t);
    t = __builtin_amdgcn_update_dpp(TNEG, v, 0x114 /*row_shr:4*/, 0xF, 0xF, false); v = max(v, t);
    t = __builtin_amdgcn_update_dpp(TNEG, v, 0x118 /*row_shr:8*/, 0xF, 0xF, false); v = max(v, t);
    t = __builtin_amdgcn_update_dpp(TNEG, v, 0x142 /*row_bcast:15*/, 0xA, 0xF, false); v = max(v, t);
    t = __builtin_amdgcn_update_dpp(TNEG, v, 0x143 /*row_bcast:31*/, 0xC, 0xF, false); v = max(v, t);
    return v;
}

struct TCand { int H, i, j; };

// TRACE: the same sweep also writes the reference's 1-byte-per-cell trace table (src/alignment/table.rs:127-142: H choice
// ZERO 0 / INS 1 / DEL 2 / DIAG 4, E bit DIAG_E 8 / INS_E 16, F bit DIAG_F 32 / DEL_F 64) with the oracle's rules
// (oracle/pmx_oracle.c: a gap "opens" when H - open > gap - ext strictly; H prefers DIAG, then DEL, then INS; SW: H <= 0 is
// ZERO).  The E bit of a cell needs the true H and E of its left neighbour: inside a lane they are at hand, for a lane's first
// column they come from lane - 1 once the row is done.  One pair's trace through this kernel: ~90 us instead of the 264 us of the
// general kernel's three 64-row bands.
template <int C, bool TRACE = false>
__global__ __launch_bounds__(64)
void pmx_table_kernel(const uint8_t *__restrict__ qbuf, const int64_t *__restrict__ qoff, int q_shared,
                      const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff, long long n,
                      const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap, int msize,
                      int mode, int sg_flags, int open, int ext,
                      const int64_t *__restrict__ tab_off /* cells before pair k's table; nullptr: one pair at 0 */,
                      int32_t *__restrict__ table, int32_t *__restrict__ row_out /* packed like the references */,
                      int32_t *__restrict__ col_out /* packed like the queries */, long long col_stride_shared,
                      pmx_record_t *__restrict__ out, int8_t *__restrict__ trace_out = nullptr /* TRACE: [qlen][rlen], one pair */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char t_lds[];
    int16_t *mat = reinterpret_cast<int16_t *>(t_lds);                 // transposed: mat[r * msize + q]
    unsigned char *rowstage = t_lds + (((size_t)msize * msize * 2 + 15) & ~(size_t)15);          // C > 4: one row of the table, 256 C bytes
    unsigned char *qs = rowstage + (C > 4 ? 256 * C : 0);                                       // mapped query symbols
    const int lane = threadIdx.x;
    const long long pair = blockIdx.x;
    const long long qb = q_shared ? 0 : qoff[pair], rb = roff[pair];
    const int ql = q_shared ? q_shared : (int)(qoff[pair + 1] - qb), rl = (int)(roff[pair + 1] - rb);
    const uint8_t *q = qbuf + qb, *r = rbuf + rb;
    for (int x = lane; x < msize * msize; x += 64) mat[(x % msize) * msize + x / msize] = gmat[x];
    for (int x = lane; x < ql; x += 64) qs[x] = gmap[q[x]];
    const bool sw = mode == PMX_MODE_SW, sg = mode == PMX_MODE_SG;
    const bool s1_end = sg && (sg_flags & PMX_SG_QE), s2_end = sg && (sg_flags & PMX_SG_DE);
    const bool col_pen = mode == PMX_MODE_NW || (sg && !(sg_flags & PMX_SG_QB));
    const bool row_pen = mode == PMX_MODE_NW || (sg && !(sg_flags & PMX_SG_DB));
    auto colB = [&](int i) -> int { return (!sw && col_pen) ? -(open + i * ext) : 0; };

    const int j0 = lane * C;
    int rbase[C], Hp[C], Fv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int j = j0 + c;
        rbase[c] = (j < rl ? (int)gmap[r[j]] : 0) * msize * 2;        // byte offset of this column's matrix row
        Hp[c] = (!sw && row_pen) ? -(open + j * ext) : 0;              // H(-1, j)
        Fv[c] = TNEG;
    }
    __syncthreads();
    const long long tab0 = tab_off ? tab_off[pair] : 0;
    int32_t *tab = table ? table + tab0 : nullptr;
    int32_t *colp = col_out ? col_out + (q_shared ? pair * col_stride_shared : qb) : nullptr;
    const int lastcol_lane = (rl - 1) / C, lastcol_c = (rl - 1) % C;
    const int decay = C * ext;
    TCand best = {TNEG, 0, 0}, bcol = {TNEG, 0, 0};

    for (int i = 0; i < ql; ++i) {
        const int qo = (int)qs[i] * 2;                                 // uniform: column of the transposed matrix
        const int hleft = colB(i);                                     // H(i, -1)
        const int dleft = i == 0 ? 0 : colB(i - 1);                    // H(i-1, -1)
        int diag = t_lane_below(Hp[C - 1], dleft);                     // H(i-1, j0-1)
        int Ht[C];
        int Tt[TRACE ? C : 1], tb[TRACE ? C : 1];                      // TRACE: T per column, trace byte under construction
        int agg = TNEG;                                                // E leaving this lane's columns if nothing came in, + (C-1) ext
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int s = *reinterpret_cast<const int16_t *>(reinterpret_cast<const unsigned char *>(mat) + rbase[c] + qo);
            const int T = diag + s;
            const int F_ext = Fv[c] - ext, F_opn = Hp[c] - open;
            int F = max(F_ext, F_opn); if (F < TNEG) F = TNEG;
            if (TRACE) { Tt[c] = T; tb[c] = F_opn > F_ext ? PARASAIL_DIAG_F : PARASAIL_DEL_F; }
            Fv[c] = F;
            diag = Hp[c];
            int h = max(T, F);
            if (sw) h = max(h, 0);
            Ht[c] = h;
            agg = max(agg, h + c * ext);                               // h - open - (C-1-c) ext, up to the constants added below
        }
        agg = agg - open - (C - 1) * ext;                              // true E into column j0 + C from inside this lane
        // E entering lane l = max over lanes l' < l of out(l') - (l-1-l') * C * ext, and the left boundary for lane 0
        int und = agg + lane * decay;                                  // un-decayed
        und = t_prefix_max(und);
        int ein = t_lane_below(und, TNEG) - (lane - 1) * decay;        // from the lanes below
        const int eb = hleft - open - lane * decay;                    // the boundary's gap, decayed over lane * C columns
        ein = max(lane == 0 ? TNEG : ein, eb);
        if (ein < TNEG) ein = TNEG;
        int e = ein;
        int rowmax = TNEG, rowj = 0;
        int e_last = TNEG;                                             // TRACE: E of the lane's last column (for the next lane's E bit)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int h = max(Ht[c], e);
            if (TRACE) {
                const int T = Tt[c], F = Fv[c];
                int hb = (T >= e && T >= F) ? PARASAIL_DIAG : (F >= e ? PARASAIL_DEL : PARASAIL_INS);
                if (sw && h <= 0) hb = PARASAIL_ZERO;
                tb[c] |= hb;
                if (c == C - 1) e_last = e;
            }
            const int e_ext = e - ext, e_opn = h - open;
            if (TRACE && c + 1 < C) tb[c + 1] |= e_opn > e_ext ? PARASAIL_DIAG_E : PARASAIL_INS_E;
            e = max(e_ext, e_opn);
            Hp[c] = h;
            if (sw && h > rowmax && j0 + c < rl) { rowmax = h; rowj = j0 + c; }
        }
        if (TRACE) {
            // E bit of the lane's first column: the left neighbour is lane - 1's last column (or the boundary column: it always opens)
            const int hl = t_lane_below(Hp[C - 1], hleft), el = t_lane_below(e_last, TNEG);
            tb[0] |= (hl - open > el - ext) ? PARASAIL_DIAG_E : PARASAIL_INS_E;
            int8_t *dst = trace_out + (long long)i * rl + j0;
#pragma unroll
            for (int c = 0; c < C; ++c) if (j0 + c < rl) dst[c] = (int8_t)tb[c];
        }
        // ---- outputs of the row ----
        if (tab && C == 4) {                                        // one 16-byte store per lane: 1 KB contiguous per wave and row
            int32_t *dst = tab + (long long)i * rl + j0;
            if (j0 + 3 < rl) { int4 v = {Hp[0], Hp[1], Hp[2], Hp[3]}; *reinterpret_cast<int4 *>(dst) = v; }
            else {
#pragma unroll
                for (int x = 0; x < 4; ++x) if (j0 + x < rl) dst[x] = Hp[x];
            }
        }
        if (tab && C > 4) {
            // C / 4 stores per lane would each touch every other cache line of the row (lane stride 4 C bytes): the row is turned in LDS
            // first, so that store x of lane l carries columns 256 x + 4 l .. + 3 -- every store is 1 KB contiguous
            int4 *stg = reinterpret_cast<int4 *>(rowstage);
#pragma unroll
            for (int c = 0; c < C; c += 4) {
                const int4 v = {Hp[c], Hp[c + 1], Hp[c + 2], Hp[c + 3]};
                const int slot16 = lane * (C / 4) + c / 4;               // 16-byte slot = 4 consecutive columns of the row
                stg[slot16 ^ ((slot16 >> 3) & 7)] = v;                    // (swizzled: the lanes' writes spread over the banks)
            }
            int32_t *dst = tab + (long long)i * rl;
#pragma unroll
            for (int x = 0; x < C / 4; ++x) {
                const int slot16 = x * 64 + lane, col = slot16 * 4;
                const int4 v = stg[slot16 ^ ((slot16 >> 3) & 7)];
                if (col + 3 < rl) *reinterpret_cast<int4 *>(dst + col) = v;
                else {
                    if (col < rl) dst[col] = v.x;
                    if (col + 1 < rl) dst[col + 1] = v.y;
                    if (col + 2 < rl) dst[col + 2] = v.z;
                }
            }
        }
        if (lane == lastcol_lane) {
            int hl = 0;
#pragma unroll
            for (int c = 0; c < C; ++c) if (c == lastcol_c) hl = Hp[c];
            if (colp) colp[i] = hl;
            if (s1_end && hl > bcol.H) { bcol.H = hl; bcol.i = i; bcol.j = rl - 1; }         // i ascends: the first maximum is kept
        }
        if (sw && (rowmax > best.H || (rowmax == best.H && rowj < best.j))) { best.H = rowmax; best.i = i; best.j = rowj; }
    }
    // last row
    if (row_out) {
        int32_t *dst = row_out + rb + j0;
#pragma unroll
        for (int c = 0; c < C; ++c) if (j0 + c < rl) dst[c] = Hp[c];
    }
    TCand brow = {TNEG, 0, 0};
    int corner = TNEG;
    if (!sw) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = j0 + c;
            if (j < rl) {
                if (s2_end && Hp[c] > brow.H) { brow.H = Hp[c]; brow.i = ql - 1; brow.j = j; }   // j ascends
                if (j == rl - 1) corner = Hp[c];
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        TCand o;
        o.H = __shfl_xor(best.H, off, 64); o.i = __shfl_xor(best.i, off, 64); o.j = __shfl_xor(best.j, off, 64);
        if (o.H > best.H || (o.H == best.H && (o.j < best.j || (o.j == best.j && o.i < best.i)))) best = o;
        o.H = __shfl_xor(brow.H, off, 64); o.i = __shfl_xor(brow.i, off, 64); o.j = __shfl_xor(brow.j, off, 64);
        if (o.H > brow.H || (o.H == brow.H && o.j < brow.j)) brow = o;
        o.H = __shfl_xor(bcol.H, off, 64); o.i = __shfl_xor(bcol.i, off, 64); o.j = __shfl_xor(bcol.j, off, 64);
        if (o.H > bcol.H) bcol = o;
        corner = max(corner, __shfl_xor(corner, off, 64));
    }
    if (lane == 0 && out) {
        pmx_record_t rec; rec.flags = 0;
        if (sw) { rec.score = best.H; rec.end_query = best.i; rec.end_ref = best.j; }
        else if (mode == PMX_MODE_NW || (!s1_end && !s2_end)) { rec.score = corner; rec.end_query = ql - 1; rec.end_ref = rl - 1; }
        else {
            TCand res = brow;
            if (s1_end && bcol.H > res.H) res = bcol;
            rec.score = res.H; rec.end_query = res.i; rec.end_ref = res.j;
        }
        out[pair] = rec;
    }
}

// 0 launched, 1 not eligible (the general kernel writes the tables), <0 HIP error
int pmx_launch_table(int mode, int sg_flags, int open, int ext, const PmxDevMatrix &m, long long n,
                     const uint8_t *qbuf, const int64_t *qoff, int q_shared, const uint8_t *rbuf, const int64_t *roff,
                     int max_qlen, int max_rlen, const int64_t *tab_off, int32_t *table, int32_t *row_out, int32_t *col_out,
                     pmx_record_t *out, hipStream_t stream, int8_t *trace)
{
    if (n <= 0) return 0;
    if (pmx_env("PMX_NO_FAST_TABLE")) return 1;
    if (open < ext || ext < 0 || m.msize > PMX_MAX_FAST_MSIZE || max_rlen > 64 * 16 || max_qlen > 100000) return 1;
    if (!tab_off && n > 1 && table) return 1;
    if (trace && n != 1) return 1;                       // the trace table form serves one pair (Aligner::align() with use_trace)
    const size_t lds = (((size_t)m.msize * m.msize * 2 + 15) & ~(size_t)15) + (size_t)max_qlen + 16 + (max_rlen > 256 ? 4096 : 0);
    if (lds > 150 * 1024) return 1;
#define LT(CC, TRF) do { const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_table_kernel<CC, TRF>)); if (rc) return rc; \
               hipLaunchKernelGGL((pmx_table_kernel<CC, TRF>), dim3((unsigned)n), dim3(64), lds, stream, qbuf, qoff, q_shared, rbuf, roff, n, \
                                  m.scores, m.mapper, m.msize, mode, sg_flags, open, ext, tab_off, table, row_out, col_out, (long long)max_qlen, out, trace); } while (0)
    if (trace) {
        if (max_rlen <= 64 * 4) LT(4, true);
        else if (max_rlen <= 64 * 8) LT(8, true);
        else LT(16, true);
    } else {
        if (max_rlen <= 64 * 4) LT(4, false);
        else if (max_rlen <= 64 * 8) LT(8, false);
        else LT(16, false);
    }
#undef LT
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
