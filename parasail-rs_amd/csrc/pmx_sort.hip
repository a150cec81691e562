// pmx_sort.hip -- length-sorted processing order for ragged batches (BASELINE config 5: references of
// 0.5-5 kbp against one query).  A slot runs for max(rlen A, rlen B) steps and a wave for the maximum
// over its slots, so pairs of similar length are grouped: keys = reference lengths, values = pair
// indices, rocPRIM radix sort (descending: the longest pairs start first, which also trims the tail
// of the launch).  Kernels read pair perm[position] and write their record to out[perm[position]],
// so results stay in input order.
#include <cstring>
#include <cstdlib>
#include "pmx_common.h"
#include <rocprim/rocprim.hpp>

__global__ void pmx_len_keys_kernel(const int64_t *roff, long long n, unsigned *keys, unsigned *vals)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = (unsigned)(roff[i + 1] - roff[i]); vals[i] = (unsigned)i; }
}

// scratch layout (caller provides `scratch` of pmx_sort_scratch_bytes(n) bytes):
//   [keys_in n][keys_out n][vals_in n][perm n][rocprim temp]
size_t pmx_sort_scratch_bytes(long long n)
{
    size_t temp = 0;
    unsigned *nul = nullptr;
    (void)rocprim::radix_sort_pairs_desc(nullptr, temp, nul, nul, nul, nul, (size_t)n, 0, 32, nullptr);
    return (size_t)n * 4 * sizeof(unsigned) + ((temp + 255) & ~(size_t)255) + 256;
}

int pmx_build_length_perm(const int64_t *d_roff, long long n, void *scratch, const unsigned **perm_out, hipStream_t stream)
{
    if (n <= 0 || n >= (1LL << 32)) return 1;
    unsigned *keys_in = (unsigned *)scratch, *keys_out = keys_in + n, *vals_in = keys_out + n, *perm = vals_in + n;
    void *temp = (void *)(((uintptr_t)(perm + n) + 255) & ~(uintptr_t)255);
    size_t temp_bytes = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, temp_bytes, keys_in, keys_out, vals_in, perm, (size_t)n, 0, 32, stream);
    hipLaunchKernelGGL(pmx_len_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_roff, n, keys_in, vals_in);
    hipError_t e = rocprim::radix_sort_pairs_desc(temp, temp_bytes, keys_in, keys_out, vals_in, perm, (size_t)n, 0, 32, stream);
    if (e != hipSuccess) return -(int)e;
    *perm_out = perm;
    return 0;
}

// ---- banded batches: processing order by the number of anti-diagonal steps of a pair's band -------------------------------------------
// The packed banded kernel (pmx_banded.hip) runs two pairs per lane group for max(steps A, steps B): neighbours in the processing
// order should need about the same number of steps.  key = steps of the band inside the matrix (the kernel's own formula).
__global__ void pmx_band_keys_kernel(const int64_t *qoff, int q_shared, const int64_t *roff, const int32_t *diag, int band, long long n,
                                     unsigned *keys, unsigned *vals, int by_entry_row)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int ql = q_shared ? q_shared : (int)(qoff[k + 1] - qoff[k]), rl = (int)(roff[k + 1] - roff[k]);
    const int d0 = diag ? diag[k] : 0, dlo = d0 - band, dhi = d0 + band;
    int s_first = 0;
    if (dlo > 0) s_first = dlo; else if (dhi < 0) s_first = -dhi;
    int i1 = ql - 1, j1 = rl - 1, s_last = -1;
    if (j1 - i1 > dhi) j1 = i1 + dhi; else if (j1 - i1 < dlo) i1 = j1 - dlo;
    if (i1 >= 0 && j1 >= 0) s_last = i1 + j1;
    if (dlo > rl - 1 || dhi < -(ql - 1)) s_last = -1;
    int ns = s_last - s_first + 1; if (ns < 0) ns = 0;
    // by_entry_row (the kernel's shared-row form: both pairs of a lane group start on one query row): first by the row where the
    // band enters the matrix, in buckets of 16 rows, then by steps
    const int row = dhi < 0 ? -dhi : 0;
    keys[k] = by_entry_row ? ((unsigned)min(row >> 4, 2047) << 20) | (unsigned)min(ns, (1 << 20) - 1) : (unsigned)ns;
    vals[k] = (unsigned)k;
}
int pmx_build_band_perm(const int64_t *d_qoff, int q_shared, const int64_t *d_roff, const int32_t *d_diag, int band, long long n,
                        void *scratch, const unsigned **perm_out, hipStream_t stream, bool by_entry_row)
{
    if (n <= 0 || n >= (1LL << 32)) return 1;
    unsigned *keys_in = (unsigned *)scratch, *keys_out = keys_in + n, *vals_in = keys_out + n, *perm = vals_in + n;
    void *temp = (void *)(((uintptr_t)(perm + n) + 255) & ~(uintptr_t)255);
    size_t temp_bytes = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, temp_bytes, keys_in, keys_out, vals_in, perm, (size_t)n, 0, 32, stream);
    hipLaunchKernelGGL(pmx_band_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_qoff, q_shared, d_roff, d_diag, band, n,
                       keys_in, vals_in, by_entry_row ? 1 : 0);
    hipError_t e = rocprim::radix_sort_pairs_desc(temp, temp_bytes, keys_in, keys_out, vals_in, perm, (size_t)n, 0, 32, stream);
    if (e != hipSuccess) return -(int)e;
    *perm_out = perm;
    return 0;
}

// ---- device CIGAR entry: exclusive scan of per-pair text lengths (int32) into int64 offsets ----
struct PmxWidenI32 { __device__ __host__ int64_t operator()(int32_t v) const { return (int64_t)v; } };

size_t pmx_text_scan_scratch_bytes(long long n)
{
    size_t temp = 0;
    auto in = rocprim::make_transform_iterator((const int32_t *)nullptr, PmxWidenI32());
    (void)rocprim::exclusive_scan(nullptr, temp, in, (int64_t *)nullptr, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(), nullptr);
    return temp + 256;
}

int pmx_launch_text_offsets(const int32_t *textlen, long long n, int64_t *text_off, void *scratch, size_t scratch_bytes, hipStream_t stream)
{
    auto in = rocprim::make_transform_iterator(textlen, PmxWidenI32());
    size_t temp = scratch_bytes;
    const hipError_t e = rocprim::exclusive_scan(scratch, temp, in, text_off, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(), stream);
    return e == hipSuccess ? 0 : -(int)e;
}


// ---- 2-bit packed DNA input: base b of `in` (byte b / 4, bits 2 (b % 4)) -> letter code -> ASCII letter in `out[b]`, b in [lo, hi) ----
__global__ void pmx_unpack2_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, long long lo4, long long hi, uint32_t letters)
{
    const long long b = (lo4 + (long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;      // four bases = one packed byte -> one dword
    if (b >= hi) return;
    const uint32_t p = in[b >> 2];
    const uint32_t w = ((letters >> (8 * (p & 3))) & 0xFF) | (((letters >> (8 * ((p >> 2) & 3))) & 0xFF) << 8) |
                       (((letters >> (8 * ((p >> 4) & 3))) & 0xFF) << 16) | (((letters >> (8 * ((p >> 6) & 3))) & 0xFF) << 24);
    *reinterpret_cast<uint32_t *>(out + b) = w;        // (bases of the neighbouring slices inside the same dword get the same values again)
}
int pmx_launch_unpack2(const uint8_t *in, uint8_t *out, long long lo, long long hi, uint32_t letters, hipStream_t stream)
{
    if (hi <= lo) return 0;
    const long long lo4 = lo / 4, n4 = (hi + 3) / 4 - lo4;
    hipLaunchKernelGGL(pmx_unpack2_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, in, out, lo4, hi, letters);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}
