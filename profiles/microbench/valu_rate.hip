// valu_rate.hip -- instruction-rate microbenchmark for the VALU roofline of the DP kernels.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run on the GPU box.
// For each candidate instruction: 8 independent chains, 64 instructions per loop iteration,
// waves_per_simd in {1,2,4,8}; prints wave64 instructions per cycle per SIMD (cycle = s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define ITER 2048

#define DEFINE_KERNEL(NAME, ASMSTR)                                                         \
__global__ __launch_bounds__(64) void k_##NAME(int *out, unsigned long long *cyc, int seed) \
{                                                                                           \
    int a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    int b = seed * 31 + 7, c = seed * 17 + 3;                                               \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
    for (int i = 0; i < ITER; ++i) {                                                        \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                     \
            asm volatile(ASMSTR : "+v"(a0) : "v"(b), "v"(c));                               \
            asm volatile(ASMSTR : "+v"(a1) : "v"(b), "v"(c));                               \
            asm volatile(ASMSTR : "+v"(a2) : "v"(b), "v"(c));                               \
            asm volatile(ASMSTR : "+v"(a3) : "v"(b), "v"(c));                               \
            asm volatile(ASMSTR : "+v"(a4) : "v"(b), "v"(c));                               \
            asm volatile(ASMSTR : "+v"(a5) : "v"(b), "v"(c));                               \
            asm volatile(ASMSTR : "+v"(a6) : "v"(b), "v"(c));                               \
            asm volatile(ASMSTR : "+v"(a7) : "v"(b), "v"(c));                               \
        }                                                                                   \
    }                                                                                       \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;            \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                        \
}

DEFINE_KERNEL(add_u32,      "v_add_u32 %0, %0, %1")
DEFINE_KERNEL(max_i32,      "v_max_i32 %0, %0, %1")
DEFINE_KERNEL(max3_i32,     "v_max3_i32 %0, %0, %1, %2")
DEFINE_KERNEL(pk_max_i16,   "v_pk_max_i16 %0, %0, %1")
DEFINE_KERNEL(pk_add_i16c,  "v_pk_add_i16 %0, %0, %1 clamp")
DEFINE_KERNEL(pk_sub_i16c,  "v_pk_sub_i16 %0, %0, %1 clamp")
DEFINE_KERNEL(pk_add_u16,   "v_pk_add_u16 %0, %0, %1")
DEFINE_KERNEL(perm_b32,     "v_perm_b32 %0, %0, %1, %2")
DEFINE_KERNEL(bfi_b32,      "v_bfi_b32 %0, %1, %0, %2")
DEFINE_KERNEL(pk_max_f16,   "v_pk_max_f16 %0, %0, %1")
DEFINE_KERNEL(pk_add_f16,   "v_pk_add_f16 %0, %0, %1")
DEFINE_KERNEL(pk_max3_f16,  "v_pk_maximum3_f16 %0, %0, %1, %2")
DEFINE_KERNEL(mov_dpp,      "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(max_i16,      "v_max_i16 %0, %0, %1")
DEFINE_KERNEL(add_f32,      "v_add_f32 %0, %0, %1")
DEFINE_KERNEL(and_or_b32,   "v_and_or_b32 %0, %0, %1, %2")
DEFINE_KERNEL(cndmask,      "v_cndmask_b32 %0, %0, %1, vcc")
DEFINE_KERNEL(add3_u32,     "v_add3_u32 %0, %0, %1, %2")
DEFINE_KERNEL(pk_mad_i16,   "v_pk_mad_i16 %0, %0, %1, %2")
DEFINE_KERNEL(sad_u16,      "v_sad_u16 %0, %0, %1, %2")


#define DEFINE_MIX(NAME, ASM_A, ASM_B)                                                      \
__global__ __launch_bounds__(64) void k_##NAME(int *out, unsigned long long *cyc, int seed) \
{                                                                                           \
    int a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    int b = seed * 31 + 7, c = seed * 17 + 3;                                               \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
    for (int i = 0; i < ITER; ++i) {                                                        \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                     \
            asm volatile(ASM_A : "+v"(a0) : "v"(b), "v"(c));                                \
            asm volatile(ASM_B : "+v"(a1) : "v"(b), "v"(c));                                \
            asm volatile(ASM_A : "+v"(a2) : "v"(b), "v"(c));                                \
            asm volatile(ASM_B : "+v"(a3) : "v"(b), "v"(c));                                \
            asm volatile(ASM_A : "+v"(a4) : "v"(b), "v"(c));                                \
            asm volatile(ASM_B : "+v"(a5) : "v"(b), "v"(c));                                \
            asm volatile(ASM_A : "+v"(a6) : "v"(b), "v"(c));                                \
            asm volatile(ASM_B : "+v"(a7) : "v"(b), "v"(c));                                \
        }                                                                                   \
    }                                                                                       \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;            \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                        \
}
DEFINE_MIX(mix_max3_sub, "v_pk_maximum3_f16 %0, %0, %1, %2", "v_sub_u32 %0, %0, %1")
DEFINE_MIX(mix_pkmax_add, "v_pk_max_i16 %0, %0, %1", "v_add_u32 %0, %0, %1")
DEFINE_MIX(mix_sub_add, "v_sub_u32 %0, %0, %1", "v_add_u32 %0, %0, %1")
DEFINE_MIX(mix_sub_xor, "v_sub_u32 %0, %0, %1", "v_xor_b32 %0, %0, %1")
DEFINE_KERNEL(sub_u32,      "v_sub_u32 %0, %0, %1")
DEFINE_KERNEL(subrev_u32,   "v_subrev_u32 %0, %1, %0")
DEFINE_KERNEL(xor_b32,      "v_xor_b32 %0, %0, %1")
DEFINE_KERNEL(max_u16,      "v_max_u16 %0, %0, %1")
DEFINE_KERNEL(lshl,         "v_lshlrev_b32 %0, 1, %0")

typedef void (*kfn)(int *, unsigned long long *, int);
struct Entry { const char *name; kfn f; };

int main()
{
    Entry tests[] = {
        {"v_add_u32", k_add_u32}, {"v_max_i32", k_max_i32}, {"v_max3_i32", k_max3_i32},
        {"v_pk_max_i16", k_pk_max_i16}, {"v_pk_add_i16 clamp", k_pk_add_i16c}, {"v_pk_sub_i16 clamp", k_pk_sub_i16c},
        {"v_pk_add_u16", k_pk_add_u16}, {"v_perm_b32", k_perm_b32}, {"v_bfi_b32", k_bfi_b32},
        {"v_pk_max_f16", k_pk_max_f16}, {"v_pk_add_f16", k_pk_add_f16}, {"v_pk_maximum3_f16", k_pk_max3_f16},
        {"v_mov_b32_dpp row_shr:1", k_mov_dpp}, {"v_max_i16", k_max_i16}, {"v_add_f32", k_add_f32},
        {"v_add3_u32", k_add3_u32}, {"v_sub_u32", k_sub_u32}, {"v_subrev_u32", k_subrev_u32}, {"v_xor_b32", k_xor_b32}, {"v_max_u16", k_max_u16}, {"v_lshlrev_b32", k_lshl}, {"mix max3_f16 + sub_u32", k_mix_max3_sub}, {"mix pk_max_i16 + add_u32", k_mix_pkmax_add}, {"mix sub_u32 + add_u32", k_mix_sub_add}, {"mix sub_u32 + xor", k_mix_sub_xor}, {"v_pk_mad_i16", k_pk_mad_i16}, {"v_sad_u16", k_sad_u16},
    };
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    int *out; unsigned long long *cyc;
    const int maxblocks = cus * 4 * 8;
    hipMalloc(&out, sizeof(int) * 64 * maxblocks); hipMalloc(&cyc, sizeof(unsigned long long) * maxblocks);
    std::vector<unsigned long long> h(maxblocks);
    printf("%-28s %10s %10s %10s %10s   (wave64 instr / cycle / SIMD; wall-derived GHz in brackets)\n", "instruction", "1w/SIMD", "2w/SIMD", "4w/SIMD", "8w/SIMD");
    for (auto &t : tests) {
        printf("%-28s", t.name);
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * 4 * wps;
            hipLaunchKernelGGL(t.f, dim3(blocks), dim3(64), 0, 0, out, cyc, 1);   // warm
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(t.f, dim3(blocks), dim3(64), 0, 0, out, cyc, 2);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
            double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
            // s_memtime ticks at a constant 100 MHz on gfx9; convert with the wall time instead:
            const double instr_per_wave = (double)ITER * 64;
            const double total_instr_per_simd = instr_per_wave * wps;
            const double sec = ms * 1e-3;
            printf(" %6.3f/ns", total_instr_per_simd / (sec * 1e9));
            (void)mean;
        }
        printf("\n");
    }
    printf("(instr/ns/SIMD: divide by the GHz the chip holds to get instr/cycle; 1.2 = one wave64 op per 2 cycles at 2.4 GHz)\n");
    return 0;
}
