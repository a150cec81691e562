// pmx_sw16.hip -- the hot kernel: local (Smith-Waterman) affine-gap alignment, score and
// end positions, many independent pairs per launch.  gfx950 only.
//
// Replaces what `Aligner::align()` reaches for the dispatch names `sw_striped_{8,16,32,64,sat}`
// (and, with traceback, `sw_trace_striped_*` in batches) (name grammar /root/reference/src/aligner/mod.rs:319-329, call site
// :411-422): the parasail CPU kernel fills the DP matrix column by column with 8/16 SIMD
// lanes striped over the query.  Here the mapping is re-designed for a 64-lane wavefront:
//
//   * A group of G adjacent lanes owns one *slot*.  A slot carries TWO pairs at once: every
//     32-bit register holds pair A's value in its low int16 and pair B's in its high int16, so
//     every v_pk_{add,sub,max}_i16 updates two DP cells.
//   * Lane g of the group keeps R consecutive query rows [g*R, g*R+R) of both pairs entirely in
//     VGPRs (H of the previous column, E) and walks the reference left to right, one column per
//     step, one step behind lane g-1 (a systolic array over the lanes).  The F chain runs down
//     the R rows inside the lane; the last row's H and F go to lane g+1 through one DPP
//     row_shr:1 / wave_shr:1 move each per step.  There is no lazy-F loop and no barrier.
//   * The per-pair query profile (score of every query row against every reference symbol,
//     int16) is built in LDS once; per step a lane fetches the R scores of its rows for the
//     current reference symbol of pair A and of pair B and interleaves them with v_perm_b32.
//   * Arithmetic variants (template parameter VAR, chosen by the host, all bit-identical in result):
//       0  saturating int16 in an offset domain (value - 32768): the clamp of v_pk_add_i16 is the zero
//          floor of local alignment, and int16 overflow (score > 32767) is detected exactly, like the
//          reference's `is_saturated` (src/alignment/mod.rs:436-440);
//       1  + v_pk_maximum3_f16 as an exact integer max3 on biased values;
//       2  + 32-bit VOP2 add/sub (profile carries score + open, strips carry H - open);
//       3  + one-byte profile entries;
//       4/5  + column-skewed values (no subtract for the E extension), int16 / byte profile;
//       6  + perm table: no LDS profile at all for alphabets of <= 4 letters (the hot configuration);
//       7  = 5 + packed 4-bit traceback output for batch CIGARs.
//     DESIGN.md section 2.1 has the derivations; the comments at each `constexpr bool` below the details.
//   * End position = first maximum in column-major order (smallest end_ref, then smallest
//     end_query): each lane keeps its running best, the column where it was first reached and a
//     copy of its R-row H strip at that column (v_bfi_b32 under a per-half mask); a group
//     reduction at the end picks (score desc, column asc, row asc).
//
// Padding never needs masks: reference positions outside [0, rlen) use a pad symbol whose
// profile row is -32768 (H falls to zero, gaps only decay), padded query rows score 0; neither
// can strictly exceed the true maximum nor precede its first occurrence in column-major order.
#include "pmx_common.h"
#include "pmx_switches.h"
#include <cstdlib>

typedef short v2s __attribute__((ext_vector_type(2)));

#define PK(x)  __builtin_bit_cast(v2s, (int)(x))
#define I32(x) __builtin_bit_cast(int, (x))

__device__ __forceinline__ v2s pk_adds(v2s a, v2s b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ v2s pk_subs(v2s a, v2s b) { return __builtin_elementwise_sub_sat(a, b); }
__device__ __forceinline__ v2s pk_max(v2s a, v2s b) { return __builtin_elementwise_max(a, b); }

// value of lane-1 inside a G-lane group; lane 0 of the group receives `neutral`.
// IL (G == 8 only): two groups share a DPP row of 16 lanes, interleaved (lane = 2 g + (slot & 1) + 16 (slot >> 1)).
// row_shr:2 then moves every group up by one lane, and the row's first two lanes -- lane 0 of both groups --
// have no source and keep `neutral`: no select is needed.
template <int G, bool IL = false>
__device__ __forceinline__ int group_shift_up(int x, int neutral, int g)
{
    if (G == 1) return neutral;
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

#define NEG16 ((short)-32768)
#define M3_BIAS 2048
#define M3_BIAS2 ((M3_BIAS << 16) | M3_BIAS)
#define M3_LIMIT(maxs) (31744 - ((maxs) > 0 ? (maxs) : 0))     // a best at or above this may have left the exact range

typedef unsigned short v2us __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s pk_subus(v2s a, v2s b)      // v_pk_sub_u16 clamp: saturates at 0
{
    return __builtin_bit_cast(v2s, __builtin_elementwise_sub_sat(__builtin_bit_cast(v2us, a), __builtin_bit_cast(v2us, b)));
}
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s pk_max3f(v2s a, v2s b, v2s c)   // integer max3 on {0} U [1024, 31743] patterns
{
    // fmaximum(fmaximum(a, b), c) on v2f16 selects v_pk_maximum3_f16 on gfx950.  A builtin rather
    // than inline asm: the hazard recognizer pads every inline-asm result with an s_nop.
    const v2h r = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(v2h, a), __builtin_bit_cast(v2h, b)),
                                                __builtin_bit_cast(v2h, c));
    return __builtin_bit_cast(v2s, r);
}
#ifndef PMX_SHARE_PERIOD
#define PMX_SHARE_PERIOD 16     // steps between two exchanges of the group's score bound (a power of two; see share_bound)
#endif
#define FLOOR2 0x80008000   // both halves = -32768 = "zero" of the offset domain

template <int G, int R, int VAR>
__global__ __launch_bounds__(64, (VAR == 6 && R <= 20) ? 4 : 1)
void pmx_sw16_kernel(const uint8_t *__restrict__ qbuf, const int64_t *__restrict__ qoff,
                     const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff,
                     long long n, const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap,
                     int msize, int open, int ext, int RP /* rsym stride, bytes */,
                     int q_shared /* > 0: every pair uses qbuf[0..q_shared) */,
                     int limit /* M3 only: biased scores at or above this are flagged for a re-run (SK: growth already taken off) */,
                     const unsigned *__restrict__ perm,
                     const int *__restrict__ n_dev /* != nullptr: the pair count is read from the device (retry launches) */,
                     unsigned *__restrict__ retry_list, int *__restrict__ retry_count /* PT: pairs to redo */,
                     int sat_above /* scores above this are flagged PMX_FLAG_SATURATED (INT_MAX: never) */,
                     uint32_t *__restrict__ tbuf, int Tmax /* VAR 7: 4-bit traceback cells, layout of pmx_nwsg16v_kernel<..,true> */,
                     pmx_record_t *__restrict__ out)
{
    if (n_dev) n = *n_dev;
    if ((long long)blockIdx.x * (2 * (64 / G)) >= n) return;
    constexpr bool M3 = VAR >= 1;      // biased unsigned lanes, v_pk_maximum3_f16 as integer max3
    constexpr bool V2 = VAR >= 2;      // + full-rate 32-bit VOP2 add/sub on packed lanes (no cross-half carry)
    constexpr bool PT = VAR == 6;      // + no LDS profile at all: alphabets of <= 4 letters (+ wildcard) look the score up with the
                                       //   v_perm itself -- table = the 4 scores of this step's reference symbol (one dword per pair),
                                       //   selector = the lane's query letters (one VGPR per row).  Query wildcards cannot be
                                       //   expressed: such pairs are flagged PMX_FLAG_RETRY16 and redone with the LDS profile.
    constexpr bool TR = VAR == 7;      // VAR 5 + packed 4-bit traceback output (see pmx_nwsg16.hip: same bits, same layout; rows top-aligned here)
    static_assert(!TR || R == 16, "trace: four packed planes of 4 rows");
    constexpr bool FETCH = VAR == 8;   // VAR 5 + reference symbols fetched from HBM two steps ahead instead of staged in LDS (long references)
    constexpr bool U8 = VAR == 3 || VAR == 5 || PT || TR || FETCH;   // + one-byte profile entries (score + open fits 0..255): half the LDS, same v_perm count
    constexpr bool SK = VAR >= 4;      // + column-skewed values (everything in column j carries +(j+G)*ext): E needs no subtract
    constexpr int EB = U8 ? 1 : 2;     // bytes per profile entry
    constexpr int WR = U8 ? 4 : 2;     // rows per loaded dword
    constexpr int RS = (R + WR - 1) / WR * WR;   // profile rows reserved per lane (whole dwords); rows R..RS-1 are unused
    constexpr int QP = G * RS;           // profile rows per pair (lane l, row k sits at l * RS + k)
    constexpr int QP2 = QP / 2;          // dwords per profile row
    constexpr int SLOTS = 64 / G;
    constexpr int NP = 2 * SLOTS;        // pairs per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int lane = threadIdx.x;
    constexpr bool IL = G == 8 && VAR != 7;     // interleaved 8-lane groups (see group_shift_up); the trace layout keeps plain groups
    const int g = IL ? (lane % 16) / 2 : lane % G;
    const int slot = IL ? (lane / 16) * 2 + (lane & 1) : lane / G;
    const int PROF_STRIDE = PT ? 0 : msize * QP * EB;   // bytes per pair

    // LDS carve: [prof NP][shared pad row QP*2][rsym NP*RP][mat msize*msize*2][map 256][pair table NP*4 ints]
    // The pad row sits right behind the last pair's profile; pair p reaches it with the symbol
    // value (NP - p) * msize, so no per-pair copy is needed.
    int16_t *prof = reinterpret_cast<int16_t *>(lds);
    unsigned char *rsym = lds + (PT ? 0 : NP * PROF_STRIDE + QP * EB);
    int16_t *mat = reinterpret_cast<int16_t *>(rsym + (FETCH ? 0 : NP * RP));
    unsigned char *map = reinterpret_cast<unsigned char *>(mat + msize * msize);
    long long *ptab = reinterpret_cast<long long *>(map + 256 + ((8 - ((msize * msize * 2) & 7)) & 7));   // per pair: q offset, qlen, r offset, rlen, pair index

    int *tabs = reinterpret_cast<int *>(ptab + 5 * NP);      // PT: per reference symbol the 4 query-letter scores (+open), entry msize = pad = 0
    constexpr int QS = (G * R + 3) / 4 * 4;
    unsigned char *qsym = reinterpret_cast<unsigned char *>(tabs + 8);   // PT: mapped query letters, QS bytes per pair (0xFF below the query)

    const long long pair0 = (long long)blockIdx.x * NP;

    // ---- stage matrix, mapper and the per-pair (offset, length) table -------------------
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

    // ---- reference symbols (with G-1 pad symbols on both sides) -------------------------
    int max_rlen = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) max_rlen = max(max_rlen, (int)ptab[5 * p + 3]);
    // Pairs in batches of UB, lanes over positions: UB independent global loads per lane are in
    // flight before anything consumes them (the prologue is latency-bound otherwise), and no
    // index needs a division.
    constexpr int UB = NP < 8 ? NP : 8;
    for (int p0 = 0; p0 < (FETCH ? 0 : NP); p0 += UB) {
        for (int j0 = 0; j0 < RP; j0 += 64) {
            const int j = j0 + lane, jr = j - (G - 1);
            unsigned char raw[UB]; bool ok[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int p = p0 + u;
                ok[u] = jr >= 0 && jr < (int)ptab[5 * p + 3];
                raw[u] = ok[u] ? rbase[ptab[5 * p + 2] + jr] : (unsigned char)0;
            }
            if (j < RP) {
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int p = p0 + u;
                    rsym[p * RP + j] = ok[u] ? map[raw[u]] : (unsigned char)(PT ? msize : (NP - p) * msize);
                }
            }
        }
    }

    if (PT) {          // query letters, staged the same way (coalesced loads; the lanes pick their rows up from LDS)
        for (int p0 = 0; p0 < NP; p0 += UB) {
            for (int j0 = 0; j0 < QS; j0 += 64) {
                const int j = j0 + lane;
                unsigned char raw[UB]; bool ok[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int p = p0 + u;
                    ok[u] = j < (int)ptab[5 * p + 1];
                    raw[u] = ok[u] ? qbase[ptab[5 * p + 0] + j] : (unsigned char)0;
                }
                if (j < QS) {
#pragma unroll
                    for (int u = 0; u < UB; ++u) qsym[(p0 + u) * QS + j] = ok[u] ? map[raw[u]] : (unsigned char)0xFF;
                }
            }
        }
    }

    // ---- query profiles: one (pair, row pair) item per lane and iteration -----------------
    constexpr int QITEMS = PT ? 0 : (NP * QP2 + 63) / 64;          // items per lane (compile time)
    constexpr int QB = QITEMS < 5 ? (QITEMS < 1 ? 1 : QITEMS) : 5;
    for (int it0 = 0; it0 < QITEMS; it0 += QB) {
        unsigned char r0[QB], r1[QB]; bool v0[QB], v1[QB];
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const int item = (it0 + u) * 64 + lane;
            const int p = min(item / QP2, NP - 1), rp = item - p * QP2;
            const int ql = (int)ptab[5 * p + 1];
            const uint8_t *qp = qbase + ptab[5 * p + 0];
            // profile position -> query row (lane l keeps rows [l * R, l * R + R) at positions l * RS + k)
            const int p0 = 2 * rp, p1 = 2 * rp + 1;
            const int k0 = p0 % RS, k1 = p1 % RS;
            const int row0 = (p0 / RS) * R + k0, row1 = (p1 / RS) * R + k1;
            v0[u] = item < NP * QP2 && k0 < R && row0 < ql; v1[u] = item < NP * QP2 && k1 < R && row1 < ql;
            r0[u] = v0[u] ? qp[row0] : (unsigned char)0;
            r1[u] = v1[u] ? qp[row1] : (unsigned char)0;
        }
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const int item = (it0 + u) * 64 + lane;
            if (it0 + u < QITEMS && item < NP * QP2) {
                const int p = item / QP2, rp = item - p * QP2;
                const int q0 = v0[u] ? map[r0[u]] : -1;
                const int q1 = v1[u] ? map[r1[u]] : -1;
                for (int sym = 0; sym < msize; ++sym) {
                    const int s0 = ((q0 < 0) ? 0 : mat[q0 * msize + sym]) + (V2 ? open : 0);
                    const int s1 = ((q1 < 0) ? 0 : mat[q1 * msize + sym]) + (V2 ? open : 0);
                    if (U8) {
                        reinterpret_cast<unsigned short *>(lds + p * PROF_STRIDE + sym * QP)[rp] =
                            (unsigned short)((s0 & 0xFF) | ((s1 & 0xFF) << 8));
                    } else {
                        int *pp = reinterpret_cast<int *>(prof) + p * (PROF_STRIDE / 4) + rp;
                        pp[sym * QP2] = (s0 & 0xFFFF) | (s1 << 16);
                    }
                }
            }
        }
    }
    if (!PT) {
        for (int idx = lane; idx < QP * EB / 4; idx += 64)
            reinterpret_cast<int *>(lds + NP * PROF_STRIDE)[idx] = V2 ? 0 : FLOOR2;
    }
    // PT: score tables and the per-row selectors (byte 0: pair A's letter 0..3 -> table A = v_perm source bytes 0..3,
    // byte 2: 4 + pair B's letter -> table B = bytes 4..7, bytes 1 and 3: 0x0C = constant 0; padding rows and wildcard
    // rows select the constant 0, i.e. score -open: harmless below the query, wrong for a wildcard -> the pair is flagged)
    int sel[R];
    int wild = 0;
    if (PT) {
        if (lane <= msize) {
            int v = 0;
            if (lane < msize)
                for (int k = 0; k < 4 && k < msize; ++k) v |= ((mat[k * msize + lane] + open) & 0xFF) << (8 * k);
            tabs[lane] = v;
        }
    }
    __syncthreads();
    if (PT) {
        const unsigned char *qa = qsym + (2 * slot) * QS + g * R, *qb = qa + QS;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int ca = qa[k], cb = qb[k];          // 0..3 letter, 4.. wildcard, 0xFF below the query
            wild |= ((ca >= 4 && ca != 0xFF) ? 1 : 0) | ((cb >= 4 && cb != 0xFF) ? 2 : 0);
            sel[k] = (ca < 4 ? ca : 0x0C) | 0x0C00 | ((cb < 4 ? 4 + cb : 0x0C) << 16) | 0x0C000000;
        }
        asm volatile("" : "+v"(wild));     // computed here: otherwise the compiler sinks it behind the sweep and spills the letters
    }
    __syncthreads();

    // ---- systolic sweep ---------------------------------------------------------------
    const int pA = 2 * slot, pB = 2 * slot + 1;
    const unsigned char *profA = lds + pA * PROF_STRIDE + g * (RS * EB);
    const unsigned char *profB = lds + pB * PROF_STRIDE + g * (RS * EB);
    const unsigned char *rsA = rsym + pA * RP + (G - 1) - g;
    const unsigned char *rsB = rsym + pB * RP + (G - 1) - g;
    const int SYMSTRIDE = QP * EB;

    const v2s vOpen = PK((open & 0xFFFF) | (open << 16));
    const v2s vExt = PK((ext & 0xFFFF) | (ext << 16));
    typedef unsigned short v2u __attribute__((ext_vector_type(2)));
    // "zero" of the value domain: -32768 for the saturating-int16 variant, BIAS for the max3 variant
    constexpr int ZERO2 = M3 ? M3_BIAS2 : FLOOR2;
    const v2s vZero = PK(ZERO2);

    // Two copies of the H strip: a step reads one and writes the other, so the loop-carried
    // values never have to be moved between registers.
    v2s HA[R], HB[R], E[R], Hsave[R];
    // V2: the strips hold H - open (the diagonal source), E starts at its exact value -open
    // SK: a value of column j is stored as value + (j + G) * ext.  Then E(j+1) = max(E(j) - ext, H(j) - open)
    // becomes E~(j+1) = max(E~(j), H~(j) - (open - ext)): the per-cell subtract of the E extension is gone.
    // F keeps an extra +ext ("F^"), so the same X = H~ - (open - ext) serves E, F and the strip (the next
    // column's diagonal source; the profile carries score + open as before).  Values only grow by
    // (steps + G) * ext; the host takes that growth off the re-run limit.  Needs open >= ext.
    const int skew0 = SK ? (((G - g) * ext) & 0xFFFF) * 0x00010001 : 0;      // this lane's first column is j = -g
    const v2s vC = PK(I32(vOpen) - I32(vExt));                               // open - ext, per half
    const v2s vInitH = V2 ? PK(ZERO2 - I32(vOpen) + skew0) : vZero;
#pragma unroll
    for (int k = 0; k < R; ++k) { HA[k] = vInitH; HB[k] = vInitH; E[k] = V2 ? vInitH : (M3 ? PK(0) : vZero); Hsave[k] = vZero; }
    v2s best = PK(ZERO2 + skew0 - (SK ? I32(vC) : 0));     // SK: X form
    int bestcol = g * 0x00010001;             // STEP at which the best was first exceeded (column = step - g; initially column 0)
    int vprev = 0, fake = 0;                  // see share_bound
    int Zv = ZERO2 + skew0 + I32(vExt);       // SK: "F^ = 0" of the current column; += ext per step
    const int HNEUTRAL = V2 ? ZERO2 - I32(vOpen) : ZERO2;
    // last-row H (V2: H - open) and outgoing F of the previous step; SK: what lane g+1 reads at step 0
    // belongs to ITS first column -(g+1), one ext below this lane's own column
    int Hout = SK ? Zv - I32(vExt) - I32(vOpen) : HNEUTRAL, Fout = SK ? Zv - I32(vExt) : ZERO2;
    v2s diag0 = PK(SK ? I32(vInitH) : HNEUTRAL);   // H(i0-1, j-1)   (V2: minus open)

    auto load_scores = [&](int symA, int symB, int (&wa)[RS / WR], int (&wb)[RS / WR]) {
        if (PT) { wa[0] = tabs[symA]; wb[0] = tabs[symB]; return; }
        const int *sa = reinterpret_cast<const int *>(profA + symA * SYMSTRIDE);
        const int *sb = reinterpret_cast<const int *>(profB + symB * SYMSTRIDE);
#pragma unroll
        for (int k = 0; k < RS / WR; ++k) { wa[k] = sa[k]; wb[k] = sb[k]; }
    };
    // trace records of 16 bytes per lane and step, lane-major: every lane's steps are contiguous, so what the walk reads along a row
    // or a diagonal sits in one cache line (consecutive stores of a lane fill its 128-byte lines in L2)
    const size_t t_ss = 4;
    uint32_t *tw = TR ? tbuf + ((size_t)blockIdx.x * Tmax) * 256 + (size_t)lane * Tmax * 4 : nullptr;
    auto push = [&](v2s &pl, v2s a, v2s b) {        // pl = 2 * pl + (a < b), per half
        const v2u fifteen = {15, 15};
        const int bit = I32(__builtin_bit_cast(v2s, __builtin_bit_cast(v2u, a - b) >> fifteen));
        int r;
        asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(I32(pl)), "v"(0x00020002), "v"(bit));
        pl = PK(r);
    };
    auto step = [&](const v2s (&Hold)[R], v2s (&Hnew)[R], const int (&wa)[RS / WR], const int (&wb)[RS / WR], int t) {
        const int Hin = group_shift_up<G, IL>(Hout, SK ? Zv - I32(vOpen) : HNEUTRAL, g); // H(i0-1, j)
        const int Fin = group_shift_up<G, IL>(Fout, SK ? Zv : ZERO2, g);                  // F(i0, j)
        v2s F = PK(Fin);
        v2s colmax = SK ? PK(0) : vZero;
        v2s Hcur[R];                                           // V2 only: this column's H (the strips hold H - open)
        v2s Tpre[R], Epre[R];                                  // V2 only: hoisted independent adds / subtracts
        v2s plane[TR ? R / 4 : 1];
        if (TR) {
#pragma unroll
            for (int x = 0; x < R / 4; ++x) plane[x] = PK(0);
        }
        if (V2) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const v2s s = PT ? PK(__builtin_amdgcn_perm(wb[0], wa[0], (unsigned)sel[k]))
                            : U8 ? PK(__builtin_amdgcn_perm(wb[k / 4], wa[k / 4], 0x0C000C00u | (unsigned)(k & 3) | ((4u + (unsigned)(k & 3)) << 16)))
                               : PK(__builtin_amdgcn_perm(wb[k / 2], wa[k / 2], (k & 1) ? 0x07060302 : 0x05040100));
                const v2s d = (k == 0) ? diag0 : Hold[k - 1];
                Tpre[k] = PK(I32(d) + I32(s));
                if (!SK) Epre[k] = PK(I32(E[k]) - I32(vExt));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const v2s s = PT ? PK(__builtin_amdgcn_perm(wb[0], wa[0], (unsigned)sel[k]))
                        : U8 ? PK(__builtin_amdgcn_perm(wb[k / 4], wa[k / 4], 0x0C000C00u | (unsigned)(k & 3) | ((4u + (unsigned)(k & 3)) << 16)))
                               : PK(__builtin_amdgcn_perm(wb[k / 2], wa[k / 2], (k & 1) ? 0x07060302 : 0x05040100));
            const v2s d = (k == 0) ? diag0 : Hold[k - 1];
            v2s H;
            if (SK) {
                const v2s Fe = PK(I32(F) - I32(vExt));            // F~ of this row (F^ - ext), also F^'s extension
                H = pk_max3f(Tpre[k], E[k], Fe);
                const v2s X = PK(I32(H) - I32(vC));
                if (TR) {
                    push(plane[k / 4], Tpre[k], H);      // ND
                    push(plane[k / 4], Fe, H);           // NDL
                    push(plane[k / 4], E[k], X);         // EO
                    push(plane[k / 4], Fe, X);           // FO
                }
                E[k] = pk_max3f(E[k], X, X);
                F = pk_max3f(Fe, X, PK(Zv));
                Hnew[k] = X;
                // the column maximum, the running best and the saved strip all live in the X form (H~ - (open - ext)):
                // the strip itself is what gets saved, no second copy of the column is kept in registers
                if (k & 1) colmax = pk_max3f(colmax, Hnew[k - 1], X);
                else if (k == R - 1) colmax = pk_max3f(colmax, X, best);      // odd R: the spare operand folds the running best in
            } else if (V2) {
                // Same domain as the max3 variant, but the strips carry H - open and the profile
                // carries score + open (>= 0), so add and subtract never carry or borrow across
                // the 16-bit halves and run as full-rate 32-bit VOP2 (v_add_u32 / v_sub_u32).
                H = pk_max3f(Tpre[k], E[k], F);
                const v2s Ho = PK(I32(H) - I32(vOpen));
                E[k] = pk_max3f(Epre[k], Ho, Ho);
                F = pk_max3f(PK(I32(F) - I32(vExt)), Ho, vZero);
                Hnew[k] = Ho;
                Hcur[k] = H;
                if (k & 1) colmax = pk_max3f(colmax, Hcur[k - 1], H);
                else if (k == R - 1) colmax = pk_max3f(colmax, H, H);
            } else if (M3) {
                // Biased unsigned lanes: every live value is 0 or in [1024, 31743], where the bit
                // patterns of non-negative f16 order like integers, so v_pk_maximum3_f16 is an exact
                // integer max3 (profiles/microbench/max3_f16_int.hip).  A pad score of -32768 sets the
                // sign bit: a negative f16 that loses against everything.
                const v2s Tt = PK(I32(__builtin_bit_cast(v2u, d) + __builtin_bit_cast(v2u, s)));
                H = pk_max3f(Tt, E[k], F);
                const v2s Ho = pk_subus(H, vOpen);
                E[k] = pk_max3f(pk_subus(E[k], vExt), Ho, Ho);
                F = pk_max3f(pk_subus(F, vExt), Ho, vZero);
                Hnew[k] = H;
                if (k & 1) colmax = pk_max3f(colmax, Hnew[k - 1], H);
                else if (k == R - 1) colmax = pk_max3f(colmax, H, H);
            } else {
                H = pk_adds(d, s);
                H = pk_max(H, E[k]);
                H = pk_max(H, F);
                const v2s Ho = pk_subs(H, vOpen);
                E[k] = pk_max(pk_subs(E[k], vExt), Ho);
                F = pk_max(pk_subs(F, vExt), Ho);
                Hnew[k] = H;
                colmax = pk_max(colmax, H);
            }
        }
        if (TR) {
            uint4 w;
            w.x = __builtin_amdgcn_perm(I32(plane[0]), I32(plane[1]), 0x00010405);   // A: bytes = row pairs (0,1) (2,3) (4,5) (6,7), even row in the high nibble
            w.y = __builtin_amdgcn_perm(I32(plane[2]), I32(plane[3]), 0x00010405);
            w.z = __builtin_amdgcn_perm(I32(plane[0]), I32(plane[1]), 0x02030607);   // B
            w.w = __builtin_amdgcn_perm(I32(plane[2]), I32(plane[3]), 0x02030607);
            *reinterpret_cast<uint4 *>(tw + (size_t)t * t_ss) = w;
        }
        diag0 = PK(Hin);
        Hout = I32(Hnew[R - 1]);
        Fout = I32(F);
        // end-position bookkeeping: strictly greater than the lane's best so far?
        constexpr bool FOLDED = SK && (R & 1);          // colmax already holds max(best, column maximum)
        const v2s nb = FOLDED ? colmax : M3 ? pk_max3f(best, colmax, colmax) : pk_max(best, colmax);
        int m;   // 0xFFFF in every half whose column maximum strictly exceeds the best so far
        {
            const v2s dd = FOLDED ? (best - nb) : M3 ? (best - colmax) : pk_subs(best, colmax);   // negative exactly where colmax > best
            const v2s sh = {15, 15};
            m = I32(dd >> sh);                              // v_pk_ashrrev_i16
        }
        // The strip is saved only in steps where some lane of the wave improves (a wave-uniform branch around R + 1 v_bfi_b32):
        // improvements get rare as a sweep goes on -- a lane's best is a running maximum -- so long sweeps skip most saves.
        if (__builtin_amdgcn_ballot_w64(m != 0) != 0) {
            // v_bfi_b32 d = (m & a) | (~m & b)
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(bestcol) : "v"(m), "s"((t & 0xFFFF) * 0x00010001), "v"(bestcol));   // the step index is uniform: SGPR operand
            fake &= ~m;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                int hs;
                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hs) : "v"(m), "v"(I32((V2 && !SK) ? Hcur[k] : Hnew[k])), "v"(I32(Hsave[k])));
                Hsave[k] = PK(hs);
            }
        }
        best = SK ? PK(I32(nb) + I32(vExt)) : nb;                           // SK: carried into the next column's skew
        if (SK) Zv += I32(vExt);
    };

    // software pipeline: scores of step t+1 are fetched from LDS while step t computes
    const int T = max_rlen + G - 1;                      // steps: the last lane's last real column
    int w0a[RS / WR], w0b[RS / WR], w1a[RS / WR], w1b[RS / WR];
    // FETCH: raw byte of step x (column x - g), -1 outside the reference; the pad symbol is the pair's own
    const int rlA_ = (int)ptab[5 * pA + 3], rlB_ = (int)ptab[5 * pB + 3];
    const uint8_t *refA = rbase + ptab[5 * pA + 2], *refB = rbase + ptab[5 * pB + 2];
    auto fetch = [&](int x, int &ra, int &rb) {
        const int col = x - g;
        ra = (col >= 0 && col < rlA_) ? (int)refA[col] : -1;
        rb = (col >= 0 && col < rlB_) ? (int)refB[col] : -1;
    };
    auto symA_of = [&](int raw) -> int { return raw < 0 ? (NP - pA) * msize : (int)map[raw]; };
    auto symB_of = [&](int raw) -> int { return raw < 0 ? (NP - pB) * msize : (int)map[raw]; };
    int m2a = 0, m2b = 0, m3a = 0, m3b = 0, nsA, nsB;
    if (FETCH) {
        int r0a, r0b, r1a, r1b;
        fetch(0, r0a, r0b); fetch(1, r1a, r1b); fetch(2, m2a, m2b); fetch(3, m3a, m3b);
        load_scores(symA_of(r0a), symB_of(r0b), w0a, w0b);
        nsA = symA_of(r1a); nsB = symB_of(r1b);
    } else {
        load_scores(rsA[0], rsB[0], w0a, w0b);
        nsA = rsA[1]; nsB = rsB[1];
    }
    // Every 16 steps the lanes of a group agree on a lower bound of their pair's final score -- the largest best any of them holds
    // (compared with the lane-dependent part of the skew taken off) -- and each raises its own `best` to one BELOW it: a lane
    // whose column maxima stay under the bound can no longer hold the pair's end cell, so its improvements need no strip save,
    // while a lane that reaches the bound itself (a tie, decided by column and row in the epilogue) still saves.  Lanes that
    // end with a raised, never-reached `best` carry a value below the pair's score into the final reduction and lose there.
    // Ties: a lane that only EQUALS the bound matters if its cell precedes the holder's in column-major order, which needs an
    // earlier column -- impossible once G steps have passed since the holder got there (lane g works on column step - g).  So
    // the bound of the PREVIOUS exchange (SHP >= G steps old) is applied in full, the fresh one less one.  A `best` raised this
    // way is not a score the lane has seen: `fake` marks those halves until the lane's next real improvement, and the epilogue
    // drops them (they could otherwise tie with the true end cell's score).
    constexpr int SHP = PMX_SHARE_PERIOD > G ? PMX_SHARE_PERIOD : G;
    auto share_bound = [&]() {
        int v = I32(best) - skew0;
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
            const int o = __shfl_xor(v, (IL ? 2 : 1) * off, 64);
            v = I32(pk_max3f(PK(v), PK(o), PK(o)));
        }
        const int fresh = v - 0x00010001 + skew0;
        const int old = vprev ? vprev + ((SHP * ext) & 0xFFFF) * 0x00010001 + skew0 : 0;        // (0: below every live value)
        const v2s nbest = pk_max3f(best, PK(fresh), PK(old));
        const v2s sh = {15, 15};
        fake |= I32((best - nbest) >> sh);                  // halves that were raised
        best = nbest;
        vprev = v;
    };
    for (int t = 0; t + 1 < T; t += 2) {
        if (SK && G > 1 && (t & (SHP - 1)) == 0 && t) share_bound();
        load_scores(nsA, nsB, w1a, w1b);
        if (FETCH) { nsA = symA_of(m2a); nsB = symB_of(m2b); fetch(t + 4, m2a, m2b); }
        else { nsA = rsA[t + 2]; nsB = rsB[t + 2]; }
        __builtin_amdgcn_sched_barrier(0);      // keep the LDS reads ahead of the step they overlap with
        step(HA, HB, w0a, w0b, t);
        __builtin_amdgcn_sched_barrier(0);
        load_scores(nsA, nsB, w0a, w0b);
        if (FETCH) { nsA = symA_of(m3a); nsB = symB_of(m3b); fetch(t + 5, m3a, m3b); }
        else { nsA = rsA[t + 3]; nsB = rsB[t + 3]; }
        __builtin_amdgcn_sched_barrier(0);
        step(HB, HA, w1a, w1b, t + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (T & 1) step(HA, HB, w0a, w0b, T - 1);                   // odd step count: one more (its scores are already loaded)

    // ---- per lane: first row of the saved strip that holds the best ---------------------
    unsigned long long keyA, keyB;
    {
        const int bA = (short)(I32(best) & 0xFFFF), bB = (short)(I32(best) >> 16);
        const unsigned stA = bestcol & 0xFFFF, stB = (unsigned)bestcol >> 16;      // save steps
        const unsigned cA = stA - g, cB = stB - g;                                 // columns
        // SK: `best` was carried through T - (save step) columns after the strip was saved
        const int tA = SK ? bA - (T - (int)stA) * ext : bA, tB = SK ? bB - (T - (int)stB) * ext : bB;
        int kA = 0, kB = 0;
#pragma unroll
        for (int k = R - 1; k >= 0; --k) {
            if ((short)(I32(Hsave[k]) & 0xFFFF) == tA) kA = k;
            if ((short)(I32(Hsave[k]) >> 16) == tB) kB = k;
        }
        const int unskew = SK ? (G - g + T) * ext - (open - ext) : 0;
        const unsigned sA = (unsigned)(bA - unskew - (M3 ? M3_BIAS : -32768)), sB = (unsigned)(bB - unskew - (M3 ? M3_BIAS : -32768));
        const unsigned rA = g * R + kA, rB = g * R + kB;
        keyA = ((unsigned long long)sA << 32) | ((0xFFFFu - cA) << 16) | (0xFFFFu - rA);
        keyB = ((unsigned long long)sB << 32) | ((0xFFFFu - cB) << 16) | (0xFFFFu - rB);
        if (fake & 0xFFFF) keyA = 0;            // a bound taken over from the group, never reached by this lane
        if (fake >> 16) keyB = 0;
    }
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) {
        const int lo = IL ? 2 * off : off;              // lane distance of group members `off` apart
        const unsigned long long oa = __shfl_xor(keyA, lo, 64), ob = __shfl_xor(keyB, lo, 64);
        keyA = oa > keyA ? oa : keyA;
        keyB = ob > keyB ? ob : keyB;
        if (PT) wild |= __shfl_xor(wild, lo, 64);
    }
    if (g == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long pi = ptab[5 * (2 * slot + h) + 4];
            if (pi >= 0) {
                const unsigned long long key = h ? keyB : keyA;
                pmx_record_t rec;
                rec.score = (int)(key >> 32);
                rec.end_ref = 0xFFFF - (int)((key >> 16) & 0xFFFF);
                rec.end_query = 0xFFFF - (int)(key & 0xFFFF);
                if (M3) rec.flags = (rec.score + M3_BIAS >= limit) ? PMX_FLAG_RERUN : 0;   // left the exact range: redo in 32 bits
                else rec.flags = rec.score > 32767 ? PMX_FLAG_SATURATED : 0;
                if (rec.score > sat_above) rec.flags |= PMX_FLAG_SATURATED;
                if (PT && ((wild >> h) & 1)) {                                              // wildcard in the query: redo with the LDS profile
                    rec.flags = PMX_FLAG_RETRY16;
                    retry_list[atomicAdd(retry_count, 1)] = (unsigned)pi;
                }
                out[pi] = rec;
            }
        }
    }
}

// ------------------------------------------------------------------------ host side ----

template <int G, int R, int VAR>
static int launch_one(const PmxBatch &b, const PmxDevMatrix &m, int open, int ext,
                      pmx_record_t *d_out, hipStream_t stream, const int *n_dev = nullptr, uint32_t *tbuf = nullptr, int Tmax = 0)
{
    constexpr bool PT = VAR == 6;
    constexpr int EB = (VAR == 3 || VAR == 5 || VAR == 7 || VAR == 8 || PT) ? 1 : 2, WR = 4 / EB, RS = (R + WR - 1) / WR * WR;
    constexpr int QP = G * RS, NP = 2 * (64 / G);
    if (NP * m.msize > 255) return 1;                 // per-pair pad symbol must fit a byte
    const int RP = ((b.max_rlen + 2 * (G - 1) + 4 + 7) / 4) * 4;
    const size_t lds = (PT ? (size_t)NP * ((G * R + 3) / 4 * 4) : (size_t)NP * m.msize * QP * EB + (size_t)QP * EB) + (VAR == 8 ? 0 : (size_t)NP * RP) +
                       (size_t)m.msize * m.msize * 2 + 256 + 8 + (size_t)NP * 40 + 32;
    if (lds > 160 * 1024) return 1;
    { const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_sw16_kernel<G, R, VAR>)); if (rc) return rc; }
    const long long blocks = (b.n + NP - 1) / NP;
    if (blocks <= 0) return 0;
    if (PT) {
        hipError_t e = hipMemsetAsync(b.retry_count, 0, sizeof(int), stream);
        if (e != hipSuccess) return -(int)e;
    }
    hipLaunchKernelGGL((pmx_sw16_kernel<G, R, VAR>), dim3((unsigned)blocks), dim3(64), lds, stream,
                       b.qbuf, b.qoff, b.rbuf, b.roff, (long long)b.n, m.scores, m.mapper,
                       m.msize, open, ext, RP, b.q_shared, M3_LIMIT(m.max) - (VAR >= 4 ? (b.max_rlen + 2 * G + 4) * ext : 0), b.perm,
                       n_dev, b.retry_list, b.retry_count, b.sat_above > 0 ? b.sat_above : 2147483647, tbuf, Tmax, d_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -(int)e;
    if (PT) {
        // Pairs with a wildcard in the query: same shape, LDS-profile variant, driven by the device-side
        // count (no host synchronisation; blocks beyond the count leave at once).
        PmxBatch r = b;
        r.perm = b.retry_list;
        return launch_one<G, R, 5>(r, m, open, ext, d_out, stream, b.retry_count);
    }
    return 0;
}

// Local alignment with traceback (batch CIGARs): the skewed byte-profile variant + packed trace output, shapes with
// 16 rows per lane; eligible exactly when that variant would be chosen for scores (so no re-run flag can occur).
static bool sw16_trace_ok(const PmxBatch &b, const PmxDevMatrix &m, int open, int ext)
{
    if (pmx_env("PMX_TRACE16_GEN1") || pmx_env("PMX_SW16_NO_SKEW") || pmx_env("PMX_SW16_NO_U8") || pmx_env("PMX_SW16_VARIANT")) return false;
    if (m.msize > PMX_MAX_FAST_MSIZE || open < ext || ext < 0 || open > 1024 || b.max_rlen > 30000 || b.q_shared || b.perm) return false;
    if (m.min < -1024 || m.max > 2048 || m.min + open < 0 || open + ext > 1024 || m.max + open > 255) return false;
    const long long feasible = (long long)(b.max_qlen < b.max_rlen ? b.max_qlen : b.max_rlen) * (m.max > 0 ? m.max : 0);
    return feasible + M3_BIAS < (long long)M3_LIMIT(m.max) - (long long)(b.max_rlen + 2 * 64 + 4) * ext;
}
int pmx_sw16_trace_plan(const PmxBatch &b, const PmxDevMatrix &m, int open, int ext, int *variant, int *Tmax, size_t *trace_bytes)
{
    if (!sw16_trace_ok(b, m, open, ext)) return 1;
    if (m.msize > 8 && m.msize < 32 && !pmx_env("PMX_SW16_NO_MATRIX_LOOKUP")) {     // large alphabet: the matrix-lookup kernel (no profile, 1 KB of LDS)
        int G = 0;
        for (int v = 1; v < 4 && !G; ++v) if (b.max_qlen <= (8 << v) * 16) { *variant = 4 + v; G = 8 << v; }
        if (!G) return 1;
        *Tmax = (b.max_rlen + G - 1 + 1 + 15) & ~15;    // that kernel sweeps an even number of steps; the walk reads windows of 16
        *trace_bytes = (size_t)((b.n + 2 * (64 / G) - 1) / (2 * (64 / G))) * (size_t)*Tmax * 64 * 16;
        return 0;
    }
    int G = 0;
    for (int v = 0; v < 4 && !G; ++v) {                  // the first shape that holds the query, whose per-pair pad symbols fit a
        const int g = 8 << v, np = 2 * (64 / g);         // byte and whose profiles fit the LDS (the launcher's own conditions)
        const size_t lds = (size_t)np * m.msize * g * 16 + (size_t)g * 16 + (size_t)np * (b.max_rlen + 2 * g + 12) +
                           (size_t)m.msize * m.msize * 2 + 256 + 8 + (size_t)np * 40 + 32;
        if (b.max_qlen <= g * 16 && np * m.msize <= 255 && lds <= 160 * 1024) { *variant = v; G = g; }
    }
    if (!G) return 1;
    const int NP = 2 * (64 / G);
    *Tmax = (b.max_rlen + G - 1 + 15) & ~15;        // multiple of 16: the walk reads a lane's records in windows of 16
    *trace_bytes = (size_t)((b.n + NP - 1) / NP) * (size_t)*Tmax * 64 * 16;
    return 0;
}
int pmx_launch_sw16_trace(int variant, const PmxBatch &b, const PmxDevMatrix &m, int open, int ext,
                          pmx_record_t *d_out, uint32_t *tbuf, int Tmax, hipStream_t stream)
{
    if (!sw16_trace_ok(b, m, open, ext)) return 1;
    if (variant >= 4) return pmx_launch_sw16m_trace(variant - 4, b, m, open, ext, d_out, tbuf, Tmax, stream);
    switch (variant) {
    case 0: return launch_one<8, 16, 7>(b, m, open, ext, d_out, stream, nullptr, tbuf, Tmax);
    case 1: return launch_one<16, 16, 7>(b, m, open, ext, d_out, stream, nullptr, tbuf, Tmax);
    case 2: return launch_one<32, 16, 7>(b, m, open, ext, d_out, stream, nullptr, tbuf, Tmax);
    case 3: return launch_one<64, 16, 7>(b, m, open, ext, d_out, stream, nullptr, tbuf, Tmax);
    }
    return 1;
}

int pmx_launch_sw16(const PmxBatch &b, const PmxDevMatrix &m, int open, int ext,
                    pmx_record_t *d_out, hipStream_t stream, const char **kernel_name)
{
    if (m.msize > PMX_MAX_FAST_MSIZE) return 1;
    if (open < 0 || ext < 0 || open > 32767 || ext > 32767) return 1;
    if (m.max > 32767 || m.min < -32767) return 1;
    if (b.max_rlen > 60000) return 1;                 // 16-bit column index
    const int q = b.max_qlen;
    // max3 variant: gap penalties and the most negative score must keep every live value >= 1024
    const char *force = pmx_env("PMX_SW16_VARIANT");
    int var = 0;
    if (open <= 1024 && ext <= 1024 && m.min >= -1024 && m.max <= 2048) var = 1;
    // 32-bit add/sub variant: score + open must be non-negative, and E - extend must not borrow
    if (var == 1 && m.min + open >= 0 && open + ext <= 1024) var = 2;
    if (force && atoi(force) < var) var = atoi(force);
    const bool u8ok = var == 2 && m.max + open <= 255 && !pmx_env("PMX_SW16_NO_U8");
    // column-skewed variant: values grow by (columns + 2 G + 4) * ext; use it only when no feasible
    // score can reach the correspondingly lower re-run limit
    const long long feasible = (long long)(b.max_qlen < b.max_rlen ? b.max_qlen : b.max_rlen) * (m.max > 0 ? m.max : 0);
    const bool sk = var == 2 && open >= ext && !pmx_env("PMX_SW16_NO_SKEW") &&
                    feasible + M3_BIAS < (long long)M3_LIMIT(m.max) - (long long)(b.max_rlen + 2 * 64 + 4) * ext;
    // alphabets of <= 4 letters (+ wildcard): no LDS profile, the v_perm looks the score up (see PT in the kernel)
    const bool pt = sk && u8ok && m.msize <= 5 && b.retry_list && b.retry_count && !b.q_has_wildcard && !pmx_env("PMX_SW16_NO_PERMTABLE");
    // one shared query (profile arm) with a real LDS profile: the workgroup-shared-profile kernel (pmx_sw16q.hip)
    if (b.q_shared && var == 2 && u8ok && sk && !pt) {
        const int rc = pmx_launch_sw16q(b, m, open, ext, d_out, stream, kernel_name);
        if (rc <= 0) return rc;
    }
    // per-pair queries over a large alphabet: no LDS profile at all, the scores are read from the matrix (pmx_sw16m.hip)
    if (!b.q_shared && var == 2 && u8ok && sk && !pt && (m.msize > 8 || pmx_env("PMX_SW16_MATRIX_LOOKUP")) && b.n > 2048) {
        const int rc = pmx_launch_sw16m(b, m, open, ext, d_out, stream, kernel_name);
        if (rc <= 0) return rc;
    }
    const bool longref = b.max_rlen >= 1024 && !pmx_env("PMX_SW16_NO_FETCH");     // staged references would dominate the LDS
#define TRY(GG, RR, NAME)                                                       \
    if (q <= (GG) * (RR)) {                                                     \
        constexpr int R4 = (RR);            /* byte-profile rows are reserved in whole dwords: any R */ \
        const bool u8 = u8ok;                                                   \
        int rc = pt ? launch_one<GG, RR, 6>(b, m, open, ext, d_out, stream)     \
               : (u8 && sk && longref) ? launch_one<GG, R4, 8>(b, m, open, ext, d_out, stream)  \
               : (u8 && sk) ? launch_one<GG, R4, 5>(b, m, open, ext, d_out, stream)  \
               : u8 ? launch_one<GG, R4, 3>(b, m, open, ext, d_out, stream)     \
               : (var == 2 && sk) ? launch_one<GG, RR, 4>(b, m, open, ext, d_out, stream)  \
               : var == 2 ? launch_one<GG, RR, 2>(b, m, open, ext, d_out, stream)  \
               : var == 1 ? launch_one<GG, RR, 1>(b, m, open, ext, d_out, stream)  \
                          : launch_one<GG, RR, 0>(b, m, open, ext, d_out, stream); \
        if (rc <= 0) { if (kernel_name) *kernel_name = pt ? NAME "/max3+vop2+skew+permtable" : var == 2 ? (sk ? ((u8 && longref) ? NAME "/max3+vop2+u8+skew+fetch" : NAME "/max3+vop2+skew") : NAME "/max3+vop2") : var == 1 ? NAME "/max3" : NAME; return rc; } \
    }
    // byte profile, 8 lanes per pair: half the fill/drain and per-step overhead of <16,10> at the same LDS;
    // rows per lane chosen for the common read lengths (100, 125, 150) so that few rows are padding
#define TRY8(RR)                                                                \
    if (u8ok && q <= 8 * (RR)) {                                                \
        int rc = pt ? launch_one<8, RR, 6>(b, m, open, ext, d_out, stream)      \
               : sk ? launch_one<8, RR, 5>(b, m, open, ext, d_out, stream) : launch_one<8, RR, 3>(b, m, open, ext, d_out, stream); \
        if (rc <= 0) { if (kernel_name) *kernel_name = pt ? "pmx_sw16_kernel<8," #RR ">/max3+vop2+skew+permtable" : sk ? "pmx_sw16_kernel<8," #RR ">/max3+vop2+u8+skew" : "pmx_sw16_kernel<8," #RR ">/max3+vop2+u8"; return rc; } \
    }
    // (a handful of pairs cannot fill the chip: then latency counts, and the 16-lane shape has half the work per step)
    if (b.n > 2048) { TRY8(7) TRY8(10) TRY8(13) TRY8(16) TRY8(19) TRY8(20) }      // 50 / 75 / 100 / 125 / 150 bp reads
#undef TRY8
    // One pair or a handful (Aligner::align() calls one at a time): nothing can fill the chip, the call's latency is the length of
    // ONE wave's dependent chain -- steps x rows per lane.  All 64 lanes on the pair and as few rows per lane as hold the query
    // (150 x 150: 213 steps of 3 rows instead of 165 steps of 10).
#define TRYLAT(RR)                                                              \
    if (u8ok && sk && !longref && b.n <= 64 && q <= 64 * (RR)) {                \
        const int rc = launch_one<64, RR, 5>(b, m, open, ext, d_out, stream);   \
        if (rc <= 0) { if (kernel_name) *kernel_name = "pmx_sw16_kernel<64," #RR ">/max3+vop2+u8+skew"; return rc; } \
    }
    TRYLAT(2) TRYLAT(3) TRYLAT(4) TRYLAT(8)
#undef TRYLAT
    TRY(16, 10, "pmx_sw16_kernel<16,10>")
    TRY(16, 16, "pmx_sw16_kernel<16,16>")
    TRY(32, 10, "pmx_sw16_kernel<32,10>")
    TRY(32, 16, "pmx_sw16_kernel<32,16>")
    TRY(64, 16, "pmx_sw16_kernel<64,16>")
    TRY(64, 32, "pmx_sw16_kernel<64,32>")
#undef TRY
    return 1;
}
