// pmx_sw16m.hip -- local alignment, score + end positions, PER-PAIR queries over LARGE alphabets (proteins:
// all-vs-all style batches, `sw_striped_{16,sat,32,64}` one-off arm).  gfx950 only.
//
// pmx_sw16.hip keeps one LDS query profile per pair: 24 symbols x 320 rows = 9 KB per pair, 37 KB per wave, one wave
// per SIMD.  Here there is NO profile: the substitution matrix itself (transposed, one 32-byte row per reference
// symbol, score + open as bytes) sits in LDS once per wave, every lane keeps the LDS offsets of its query letters
// in registers, and each row's score is one ds_read_u8 at (row of this step's reference symbol) + (letter offset)
// -- two byte reads and three VALU instructions per row and pair slot instead of one v_perm, in exchange for
// 1 KB of LDS per wave.  Reference symbols are fetched from HBM two steps ahead.  Mapping, arithmetic
// (column-skewed values, v_pk_maximum3_f16 as integer max3, VOP2 add/sub) and results are those of
// pmx_sw16.hip's skewed variants.
#include "pmx_common.h"
#include "pmx_switches.h"
#include <cstdlib>

typedef short m_v2s __attribute__((ext_vector_type(2)));
typedef _Float16 m_v2h __attribute__((ext_vector_type(2)));
#define M_PK(x)  __builtin_bit_cast(m_v2s, (int)(x))
#define M_I32(x) __builtin_bit_cast(int, (x))
#define M_BIASx 2048
#define M_BIAS2x ((M_BIASx << 16) | M_BIASx)
#define M_LIMITx(maxs) (31744 - ((maxs) > 0 ? (maxs) : 0))

__device__ __forceinline__ int m_max3(int a, int b, int c)
{
    const m_v2h r = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(m_v2h, a), __builtin_bit_cast(m_v2h, b)),
                                                  __builtin_bit_cast(m_v2h, c));
    return __builtin_bit_cast(int, r);
}
// value of lane-1 inside a G-lane group; lane 0 of the group receives `neutral`.
template <int G>
__device__ __forceinline__ int m_shift_up(int x, int neutral, int g)
{
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

// TR: also writes the 4-bit traceback cells (bits and layout of pmx_nwsg16v_kernel<..,true>; rows top-aligned).
template <int G, int R, bool TR>
__global__ __launch_bounds__(64)
void pmx_sw16m_kernel(const uint8_t *__restrict__ qbuf, const int64_t *__restrict__ qoff,
                      const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff,
                      long long n, const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap,
                      int msize, int open, int ext,
                      int limit /* biased scores at or above this are flagged for a re-run (skew growth already taken off) */,
                      int sat_above, const unsigned *__restrict__ perm,
                      pmx_record_t *__restrict__ out, uint32_t *__restrict__ tbuf, int Tmax)
{
    static_assert(!TR || R == 16, "trace: four packed planes of 4 rows");
    constexpr int NPW = 2 * (64 / G);           // pairs per wave = pairs per workgroup
    constexpr int MSTR = 32;                    // bytes per row of the transposed matrix
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int lane = threadIdx.x;
    const int g = lane % G, slotw = lane / G;
    const int pA = 2 * slotw, pB = pA + 1;

    // LDS: matT[(msize + 1) rows][MSTR]: matT[r][q] = score(q, r) + open for letters; column msize (rows beyond the query)
    // = open (score 0); row msize (outside the reference) = 0 (score -open) -- [map 256][ptab]
    unsigned char *matT = lds;
    unsigned char *map = lds + (msize + 1) * MSTR;
    long long *ptab = reinterpret_cast<long long *>(map + 256);       // per pair: q offset, qlen, r offset, rlen, pair index

    const long long pair0 = (long long)blockIdx.x * NPW;
    for (int i = lane; i < (msize + 1) * MSTR; i += 64) {
        const int r = i / MSTR, q = i % MSTR;
        matT[i] = (unsigned char)(r == msize ? 0 : q < msize ? gmat[q * msize + r] + open : open);
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
    const int rlA = (int)ptab[5 * pA + 3], rlB = (int)ptab[5 * pB + 3];
    const uint8_t *refA = rbuf + ptab[5 * pA + 2], *refB = rbuf + ptab[5 * pB + 2];
    // LDS offsets of this lane's query letters inside a matT row (column msize beyond the query)
    int qa[R], qb_[R];
    {
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
        }
    }
    auto fetch = [&](int x, int &ra, int &rb) {     // raw byte of step x: column x - g, -1 outside the reference
        const int col = x - g;
        ra = (col >= 0 && col < rlA) ? (int)refA[col] : -1;
        rb = (col >= 0 && col < rlB) ? (int)refB[col] : -1;
    };
    auto row_of = [&](int raw) -> int { return (raw < 0 ? msize : (int)map[raw]) * MSTR; };   // byte offset of the matT row

    auto pack2 = [](int a, int b) -> int { return (a & 0xFFFF) | (b << 16); };
    const int vOpen = pack2(open, open), vExt = pack2(ext, ext), vC = vOpen - vExt;
    const int skew0 = (((G - g) * ext) & 0xFFFF) * 0x00010001;       // this lane's first column is j = -g
    const int vInitH = M_BIAS2x - vOpen + skew0;

    int X[R], E[R], Hsave[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { X[k] = vInitH; E[k] = vInitH; Hsave[k] = M_BIAS2x; }
    int best = M_BIAS2x + skew0 - vC;            // X form
    int bestcol = g * 0x00010001;               // step of the first strict improvement (column = step - g)
    int Zv = M_BIAS2x + skew0 + vExt;            // "F^ = 0" of the current column; += ext per step
    int Hout = Zv - vExt - vOpen, Fout = Zv - vExt;
    int diag0 = vInitH;

    int w[2][R];                                // packed scores (pair A low half, pair B high half) of the next two steps
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
        typedef unsigned short u2 __attribute__((ext_vector_type(2)));
        const u2 fifteen = {15, 15};
        const int bit = M_I32(__builtin_bit_cast(m_v2s, __builtin_bit_cast(u2, M_PK(a) - M_PK(b)) >> fifteen));
        int r;
        asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(pl), "v"(0x00020002), "v"(bit));
        pl = r;
    };
    auto step = [&](int bsel, int t) {
        const int Hin = m_shift_up<G>(Hout, Zv - vOpen, g);
        int F = m_shift_up<G>(Fout, Zv, g);
        int T[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            T[k] = ((k == 0) ? diag0 : X[k - 1]) + w[bsel][k];
        }
        __builtin_amdgcn_sched_barrier(0);
        int colmax = 0;
        int plane[TR ? R / 4 : 1];
        if (TR) {
#pragma unroll
            for (int x = 0; x < R / 4; ++x) plane[x] = 0;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int Fe = F - vExt;
            const int H = m_max3(T[k], E[k], Fe);
            const int Xn = H - vC;
            if (TR) {
                push(plane[k / 4], T[k], H);         // ND
                push(plane[k / 4], Fe, H);           // NDL
                push(plane[k / 4], E[k], Xn);        // EO
                push(plane[k / 4], Fe, Xn);          // FO
            }
            E[k] = m_max3(E[k], Xn, Xn);
            F = m_max3(Fe, Xn, Zv);
            X[k] = Xn;
            if (k & 1) colmax = (k == 1) ? m_max3(X[0], Xn, Xn) : m_max3(colmax, X[k - 1], Xn);
            else if (k == R - 1) colmax = m_max3(colmax, Xn, Xn);
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
        const int nb = m_max3(best, colmax, colmax);
        const m_v2s sh = {15, 15};
        const int m = M_I32((M_PK(best) - M_PK(colmax)) >> sh);     // 0xFFFF where the column maximum strictly exceeds the best so far
        if (__builtin_amdgcn_ballot_w64(m != 0) != 0) {            // (wave-uniform: most steps of a long sweep improve no lane's best)
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(bestcol) : "v"(m), "s"((t & 0xFFFF) * 0x00010001), "v"(bestcol));
#pragma unroll
            for (int k = 0; k < R; ++k) {
                int hs;
                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hs) : "v"(m), "v"(X[k]), "v"(Hsave[k]));
                Hsave[k] = hs;
            }
        }
        best = nb + vExt;
        Zv += vExt;
    };

    int max_rlen = 0;
#pragma unroll
    for (int p = 0; p < NPW; ++p) max_rlen = max(max_rlen, (int)ptab[5 * p + 3]);
    const int T_ = (max_rlen + G - 1 + 1) & ~1;
    int r0a, r0b, r1a, r1b, m2a, m2b, m3a, m3b;
    fetch(0, r0a, r0b); fetch(1, r1a, r1b); fetch(2, m2a, m2b); fetch(3, m3a, m3b);
    load_scores(0, row_of(r0a), row_of(r0b));
    int nsA = row_of(r1a), nsB = row_of(r1b);
    for (int t = 0; t < T_; t += 2) {
        load_scores(1, nsA, nsB);
        nsA = row_of(m2a); nsB = row_of(m2b);
        fetch(t + 4, m2a, m2b);
        __builtin_amdgcn_sched_barrier(0);
        step(0, t);
        __builtin_amdgcn_sched_barrier(0);
        load_scores(0, nsA, nsB);
        nsA = row_of(m3a); nsB = row_of(m3b);
        fetch(t + 5, m3a, m3b);
        __builtin_amdgcn_sched_barrier(0);
        step(1, t + 1);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- per lane: first row of the saved strip that holds the best; then the group's winner ----
    unsigned long long keyA, keyB;
    {
        const int bA = (short)(best & 0xFFFF), bB = (short)(best >> 16);
        const unsigned stA = bestcol & 0xFFFF, stB = (unsigned)bestcol >> 16;
        const unsigned cA = stA - g, cB = stB - g;
        const int tA = bA - (T_ - (int)stA) * ext, tB = bB - (T_ - (int)stB) * ext;   // `best` was carried through the later columns
        int kA = 0, kB = 0;
#pragma unroll
        for (int k = R - 1; k >= 0; --k) {
            if ((short)(Hsave[k] & 0xFFFF) == tA) kA = k;
            if ((short)(Hsave[k] >> 16) == tB) kB = k;
        }
        const int unskew = (G - g + T_) * ext - (open - ext);
        const unsigned sA = (unsigned)(bA - unskew - M_BIASx), sB = (unsigned)(bB - unskew - M_BIASx);
        const unsigned rA = g * R + kA, rB = g * R + kB;
        keyA = ((unsigned long long)sA << 32) | ((0xFFFFu - cA) << 16) | (0xFFFFu - rA);
        keyB = ((unsigned long long)sB << 32) | ((0xFFFFu - cB) << 16) | (0xFFFFu - rB);
    }
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) {
        const unsigned long long oa = __shfl_xor(keyA, off, 64), ob = __shfl_xor(keyB, off, 64);
        keyA = oa > keyA ? oa : keyA;
        keyB = ob > keyB ? ob : keyB;
    }
    if (g == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long pi = ptab[5 * (h ? pB : pA) + 4];
            if (pi >= 0) {
                const unsigned long long key = h ? keyB : keyA;
                pmx_record_t rec;
                rec.score = (int)(key >> 32);
                rec.end_ref = 0xFFFF - (int)((key >> 16) & 0xFFFF);
                rec.end_query = 0xFFFF - (int)(key & 0xFFFF);
                rec.flags = (rec.score + M_BIASx >= limit) ? PMX_FLAG_RERUN : 0;
                if (rec.score > sat_above) rec.flags |= PMX_FLAG_SATURATED;
                out[pi] = rec;
            }
        }
    }
}

template <int G, int R, bool TR = false>
static int launch_m(const PmxBatch &b, const PmxDevMatrix &m, int open, int ext, pmx_record_t *d_out, hipStream_t stream,
                    uint32_t *tbuf = nullptr, int Tmax = 0)
{
    constexpr int NP = 2 * (64 / G);
    const size_t lds = (size_t)(m.msize + 1) * 32 + 256 + (size_t)NP * 40;
    const long long blocks = (b.n + NP - 1) / NP;
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL((pmx_sw16m_kernel<G, R, TR>), dim3((unsigned)blocks), dim3(64), lds, stream,
                       b.qbuf, b.qoff, b.rbuf, b.roff, (long long)b.n, m.scores, m.mapper, m.msize, open, ext,
                       M_LIMITx(m.max) - (b.max_rlen + 2 * G + 4) * ext, b.sat_above > 0 ? b.sat_above : 2147483647, b.perm, d_out, tbuf, Tmax);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// traceback variant, lane groups of 16 / 32 / 64 (variant 1..3 of pmx_sw16_trace_plan)
int pmx_launch_sw16m_trace(int variant, const PmxBatch &b, const PmxDevMatrix &m, int open, int ext,
                           pmx_record_t *d_out, uint32_t *tbuf, int Tmax, hipStream_t stream)
{
    switch (variant) {
    case 1: return launch_m<16, 16, true>(b, m, open, ext, d_out, stream, tbuf, Tmax);
    case 2: return launch_m<32, 16, true>(b, m, open, ext, d_out, stream, tbuf, Tmax);
    case 3: return launch_m<64, 16, true>(b, m, open, ext, d_out, stream, tbuf, Tmax);
    }
    return 1;
}

// 0 launched, 1 not eligible (the caller goes on with pmx_sw16's own variants), <0 HIP error.
// The caller has already established the conditions of the skewed byte-profile variant.
int pmx_launch_sw16m(const PmxBatch &b, const PmxDevMatrix &m, int open, int ext,
                     pmx_record_t *d_out, hipStream_t stream, const char **kernel_name)
{
    if (b.q_shared || m.msize > 31 || pmx_env("PMX_SW16_NO_MATRIX_LOOKUP")) return 1;
    const int q = b.max_qlen;
#define TRYM(GG, RR, NAME)                                                      \
    if (q <= (GG) * (RR)) {                                                     \
        int rc = launch_m<GG, RR>(b, m, open, ext, d_out, stream);             \
        if (rc <= 0) { if (kernel_name) *kernel_name = NAME; return rc; }       \
    }
    TRYM(16, 10, "pmx_sw16m_kernel<16,10>/matrix lookup")
    TRYM(16, 16, "pmx_sw16m_kernel<16,16>/matrix lookup")
    TRYM(32, 10, "pmx_sw16m_kernel<32,10>/matrix lookup")
    TRYM(32, 16, "pmx_sw16m_kernel<32,16>/matrix lookup")
    TRYM(64, 16, "pmx_sw16m_kernel<64,16>/matrix lookup")
#undef TRYM
    return 1;
}
