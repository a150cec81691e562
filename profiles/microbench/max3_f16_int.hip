// Does v_pk_maximum3_f16 act as an exact integer max3 on bit patterns {0} U [1024, 31743]?
// (non-negative f16 patterns order like unsigned integers; 1..1023 are denormals, >= 31744 Inf/NaN)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void check(unsigned long long *bad, unsigned *first)
{
    const unsigned a = blockIdx.x + 1024u;                 // 1024 .. 31743
    if (a > 31743u) return;
    const unsigned cs[6] = {0u, 1024u, 2048u, 15000u, 31743u, a};
    unsigned long long nbad = 0;
    for (unsigned b0 = threadIdx.x; b0 <= 30720u; b0 += blockDim.x) {
        const unsigned b = b0 == 30720u ? 0u : b0 + 1024u;
        for (int k = 0; k < 6; ++k) {
            const unsigned c = cs[k];
            unsigned x = a | (b << 16), y = b | (c << 16), z = c | (a << 16), r;
            asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
            const unsigned m = max(max(a, b), c);
            if ((r & 0xFFFF) != m || (r >> 16) != m) { if (!nbad) atomicCAS(first, 0u, (a << 16) | b); ++nbad; }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main()
{
    unsigned long long *bad, h = 0; unsigned *first, hf = 0;
    hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMemset(bad, 0, 8); hipMemset(first, 0, 4);
    hipLaunchKernelGGL(check, dim3(30720), dim3(256), 0, 0, bad, first);
    hipDeviceSynchronize();
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("v_pk_maximum3_f16 as integer max3 on {0} U [1024,31743]: mismatches = %llu (first a=%u b=%u)\n", h, hf >> 16, hf & 0xFFFF);
    return h != 0;
}
