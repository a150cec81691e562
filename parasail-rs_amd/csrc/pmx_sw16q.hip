// pmx_sw16q.hip -- local alignment, score + end positions, for ONE SHARED QUERY against many references
// (the profile arm, /root/reference/src/aligner/mod.rs:431-450: `sw_striped_profile_{16,sat,32,64}` -- the classic
// database search: one protein query, a reused profile, thousands of references).  gfx950 only.
//
// Same machine mapping and arithmetic as the skewed byte-profile variant of pmx_sw16.hip (two pairs per lane
// slot, column-skewed values, profile byte = score + open, v_pk_maximum3_f16 as integer max3, end position =
// first maximum in column-major order through a saved strip), specialised for the shared query:
//   * the query profile is built ONCE per 4-wave workgroup (24 symbols x 320 rows = 9 KB instead of 9 KB per
//     pair), and
//   * reference symbols are not staged in LDS: each lane fetches its next symbols from HBM two steps ahead
//     (neighbouring lanes read neighbouring bytes).  16 references of 5 kaa would otherwise take 80 KB.
// LDS drops from 57 KB per wave to 11 KB per workgroup, the occupancy from 0.5 to 4 waves per SIMD.
#include "pmx_common.h"
#include "pmx_switches.h"
#include <cstdlib>

typedef short q_v2s __attribute__((ext_vector_type(2)));
typedef _Float16 q_v2h __attribute__((ext_vector_type(2)));
#define Q_PK(x)  __builtin_bit_cast(q_v2s, (int)(x))
#define Q_I32(x) __builtin_bit_cast(int, (x))
#define Q_BIAS 2048
#define Q_BIAS2 ((Q_BIAS << 16) | Q_BIAS)
#define Q_LIMIT(maxs) (31744 - ((maxs) > 0 ? (maxs) : 0))

__device__ __forceinline__ int q_max3(int a, int b, int c)
{
    const q_v2h r = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(q_v2h, a), __builtin_bit_cast(q_v2h, b)),
                                                  __builtin_bit_cast(q_v2h, c));
    return __builtin_bit_cast(int, r);
}
// value of lane-1 inside a G-lane group; lane 0 of the group receives `neutral`.
template <int G>
__device__ __forceinline__ int q_shift_up(int x, int neutral, int g)
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

template <int G, int R, int WAVES>
__global__ __launch_bounds__(64 * WAVES)
void pmx_sw16q_kernel(const uint8_t *__restrict__ qbuf, int qlen,
                      const uint8_t *__restrict__ rbuf, const int64_t *__restrict__ roff,
                      long long n, const int16_t *__restrict__ gmat, const uint8_t *__restrict__ gmap,
                      int msize, int open, int ext,
                      int limit /* biased scores at or above this are flagged for a re-run (skew growth already taken off) */,
                      int sat_above, const unsigned *__restrict__ perm,
                      pmx_record_t *__restrict__ out)
{
    constexpr int RS = (R + 3) / 4 * 4;
    constexpr int QPS = G * RS;                 // profile bytes per symbol
    constexpr int NPW = 2 * (64 / G);           // pairs per wave
    constexpr int NP = NPW * WAVES;
    constexpr int NT = 64 * WAVES;
    constexpr int W4 = RS / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane % G, slotw = lane / G;
    const int pA = wave * NPW + 2 * slotw, pB = pA + 1;
    const int MS1 = msize + 1;                  // + the pad symbol: profile row 0 (score -open)

    unsigned char *psc = lds;                   // [MS1][QPS]
    int16_t *mat = reinterpret_cast<int16_t *>(lds + ((MS1 * QPS + 7) & ~7));
    unsigned char *map = reinterpret_cast<unsigned char *>(mat + msize * msize);
    long long *ptab = reinterpret_cast<long long *>(map + 256 + ((8 - ((msize * msize * 2) & 7)) & 7));   // per pair: r offset, rlen, pair index

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
    // the shared profile: query row i = l * R + k sits at byte l * RS + k; rows beyond the query score 0
    for (int er = tid; er < G * R; er += NT) {
        const int q0 = (er < qlen) ? (int)map[qbuf[er]] : -1;
        unsigned char *sc = psc + (er / R) * RS + er % R;
        for (int sym = 0; sym < msize; ++sym) sc[sym * QPS] = (unsigned char)(((q0 < 0) ? 0 : mat[q0 * msize + sym]) + open);
        sc[msize * QPS] = 0;
    }
    __syncthreads();

    const unsigned char *scL = psc + g * RS;
    const int rlA = (int)ptab[3 * pA + 1], rlB = (int)ptab[3 * pB + 1];
    const uint8_t *refA = rbuf + ptab[3 * pA + 0], *refB = rbuf + ptab[3 * pB + 0];
    auto fetch = [&](int x, int &ra, int &rb) {     // raw byte of step x: column x - g, -1 outside the reference
        const int col = x - g;
        ra = (col >= 0 && col < rlA) ? (int)refA[col] : -1;
        rb = (col >= 0 && col < rlB) ? (int)refB[col] : -1;
    };
    auto sym_of = [&](int raw) -> int { return raw < 0 ? msize : (int)map[raw]; };

    auto pack2 = [](int a, int b) -> int { return (a & 0xFFFF) | (b << 16); };
    const int vOpen = pack2(open, open), vExt = pack2(ext, ext), vC = vOpen - vExt;
    const int skew0 = (((G - g) * ext) & 0xFFFF) * 0x00010001;       // this lane's first column is j = -g
    const int vInitH = Q_BIAS2 - vOpen + skew0;

    int X[R], E[R], Hsave[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { X[k] = vInitH; E[k] = vInitH; Hsave[k] = Q_BIAS2; }
    int best = Q_BIAS2 + skew0 - vC;            // X form
    int bestcol = g * 0x00010001;               // step of the first strict improvement (column = step - g)
    int Zv = Q_BIAS2 + skew0 + vExt;            // "F^ = 0" of the current column; += ext per step
    int Hout = Zv - vExt - vOpen, Fout = Zv - vExt;
    int diag0 = vInitH;

    int w[2][2][W4];
    auto load_scores = [&](int bsel, int symA, int symB) {
        const int *a = reinterpret_cast<const int *>(scL + symA * QPS), *b = reinterpret_cast<const int *>(scL + symB * QPS);
#pragma unroll
        for (int x = 0; x < W4; ++x) { w[bsel][0][x] = a[x]; w[bsel][1][x] = b[x]; }
    };
    auto step = [&](int bsel, int t) {
        const int Hin = q_shift_up<G>(Hout, Zv - vOpen, g);
        int F = q_shift_up<G>(Fout, Zv, g);
        int T[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int s = __builtin_amdgcn_perm(w[bsel][1][k / 4], w[bsel][0][k / 4], 0x0C000C00u | (unsigned)(k & 3) | ((4u + (unsigned)(k & 3)) << 16));
            T[k] = ((k == 0) ? diag0 : X[k - 1]) + s;
        }
        __builtin_amdgcn_sched_barrier(0);
        int colmax = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int Fe = F - vExt;
            const int H = q_max3(T[k], E[k], Fe);
            const int Xn = H - vC;
            E[k] = q_max3(E[k], Xn, Xn);
            F = q_max3(Fe, Xn, Zv);
            X[k] = Xn;
            if (k & 1) colmax = (k == 1) ? q_max3(X[0], Xn, Xn) : q_max3(colmax, X[k - 1], Xn);
            else if (k == R - 1) colmax = q_max3(colmax, Xn, Xn);
        }
        diag0 = Hin;
        Hout = X[R - 1];
        Fout = F;
        const int nb = q_max3(best, colmax, colmax);
        const q_v2s sh = {15, 15};
        const int m = Q_I32((Q_PK(best) - Q_PK(colmax)) >> sh);     // 0xFFFF where the column maximum strictly exceeds the best so far
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
    for (int p = 0; p < NPW; ++p) max_rlen = max(max_rlen, (int)ptab[3 * (wave * NPW + p) + 1]);
    const int T_ = (max_rlen + G - 1 + 1) & ~1;
    int r0a, r0b, r1a, r1b, m2a, m2b, m3a, m3b;
    fetch(0, r0a, r0b); fetch(1, r1a, r1b); fetch(2, m2a, m2b); fetch(3, m3a, m3b);
    load_scores(0, sym_of(r0a), sym_of(r0b));
    int nsA = sym_of(r1a), nsB = sym_of(r1b);
    for (int t = 0; t < T_; t += 2) {
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
        const unsigned sA = (unsigned)(bA - unskew - Q_BIAS), sB = (unsigned)(bB - unskew - Q_BIAS);
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
            const long long pi = ptab[3 * (h ? pB : pA) + 2];
            if (pi >= 0) {
                const unsigned long long key = h ? keyB : keyA;
                pmx_record_t rec;
                rec.score = (int)(key >> 32);
                rec.end_ref = 0xFFFF - (int)((key >> 16) & 0xFFFF);
                rec.end_query = 0xFFFF - (int)(key & 0xFFFF);
                rec.flags = (rec.score + Q_BIAS >= limit) ? PMX_FLAG_RERUN : 0;
                if (rec.score > sat_above) rec.flags |= PMX_FLAG_SATURATED;
                out[pi] = rec;
            }
        }
    }
}

template <int G, int R, int WAVES>
static int launch_q(const PmxBatch &b, const PmxDevMatrix &m, int open, int ext, pmx_record_t *d_out, hipStream_t stream)
{
    constexpr int RS = (R + 3) / 4 * 4, NP = 2 * (64 / G) * WAVES;
    const size_t lds = (size_t)(m.msize + 1) * G * RS + 8 + (size_t)m.msize * m.msize * 2 + 256 + 8 + (size_t)NP * 24;
    if (lds > 160 * 1024) return 1;
    { const int rc = pmx_ensure_lds_attr(reinterpret_cast<const void *>(&pmx_sw16q_kernel<G, R, WAVES>)); if (rc) return rc; }
    const long long blocks = (b.n + NP - 1) / NP;
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL((pmx_sw16q_kernel<G, R, WAVES>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, stream,
                       b.qbuf, b.q_shared, b.rbuf, b.roff, (long long)b.n, m.scores, m.mapper, m.msize, open, ext,
                       Q_LIMIT(m.max) - (b.max_rlen + 2 * G + 4) * ext, b.sat_above > 0 ? b.sat_above : 2147483647, b.perm, d_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

// 0 launched, 1 not eligible (the caller goes on with pmx_sw16's own variants), <0 HIP error.
// The caller has already established the conditions of the skewed byte-profile variant.
int pmx_launch_sw16q(const PmxBatch &b, const PmxDevMatrix &m, int open, int ext,
                     pmx_record_t *d_out, hipStream_t stream, const char **kernel_name)
{
    if (!b.q_shared || pmx_env("PMX_SW16_NO_SHARED")) return 1;
    const int q = b.max_qlen;
#define TRYQ(GG, RR, NAME)                                                      \
    if (q <= (GG) * (RR)) {                                                     \
        int rc = launch_q<GG, RR, 4>(b, m, open, ext, d_out, stream);          \
        if (rc <= 0) { if (kernel_name) *kernel_name = NAME; return rc; }       \
    }
    TRYQ(16, 10, "pmx_sw16q_kernel<16,10>/shared profile")
    TRYQ(16, 16, "pmx_sw16q_kernel<16,16>/shared profile")
    TRYQ(32, 10, "pmx_sw16q_kernel<32,10>/shared profile")
    TRYQ(32, 16, "pmx_sw16q_kernel<32,16>/shared profile")
    TRYQ(64, 16, "pmx_sw16q_kernel<64,16>/shared profile")
    TRYQ(64, 32, "pmx_sw16q_kernel<64,32>/shared profile")
#undef TRYQ
    return 1;
}
