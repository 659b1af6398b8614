/* isa_rate_probe.hip — design probe (not product): issue cost of the VALU / LDS instructions the tree walk is made of,
 * on gfx950, at 1 / 2 / 4 / 8 waves per SIMD.  Prints shader cycles per wave-instruction per SIMD.
 * Build: hipcc --offload-arch=gfx950 -O2 tools/isa_rate_probe.hip -o gpurun_out/isa_rate_probe */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ITER 2000
#define REP8(x) x x x x x x x x

#define KERNEL64(name, INS)                                                                                            \
    __global__ __launch_bounds__(256) void k_##name(unsigned long long *out, double seed)                              \
    {                                                                                                                  \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,   \
               a7 = a0 + 7;                                                                                            \
        double b = seed * 0.5 + 1.0, c = seed * 0.25 + 0.125;                                                          \
        asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");                          \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for(int it = 0; it < ITER; it++) {                                                                             \
            asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                       \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                         : "v"(b), "v"(c)                                                                              \
                         : "vcc", "s20", "s21");                                                                                     \
            asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                       \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                         : "v"(b), "v"(c)                                                                              \
                         : "vcc", "s20", "s21");                                                                                     \
        }                                                                                                              \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        if((threadIdx.x & 63) == 0)                                                                                    \
            out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                                        \
        if(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345)                                                            \
            out[0] = 0;                                                                                                \
    }

#define KERNEL32(name, INS)                                                                                            \
    __global__ __launch_bounds__(256) void k_##name(unsigned long long *out, double seed)                              \
    {                                                                                                                  \
        float a0 = (float) seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,         \
              a6 = a0 + 6, a7 = a0 + 7;                                                                                \
        float b = (float) seed * 0.5f + 1.0f, c = (float) seed * 0.25f + 0.125f;                                       \
        asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");                          \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for(int it = 0; it < ITER; it++) {                                                                             \
            asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                       \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                         : "v"(b), "v"(c)                                                                              \
                         : "vcc", "s20", "s21");                                                                                     \
            asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                       \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                         : "v"(b), "v"(c)                                                                              \
                         : "vcc", "s20", "s21");                                                                                     \
        }                                                                                                              \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        if((threadIdx.x & 63) == 0)                                                                                    \
            out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                                        \
        if(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345f)                                                           \
            out[0] = 0;                                                                                                \
    }

/* f64 */
#define I_FMA64(r) "v_fma_f64 %" #r ", %" #r ", %8, %9\n"
#define I_ADD64(r) "v_add_f64 %" #r ", %" #r ", %8\n"
#define I_MUL64(r) "v_mul_f64 %" #r ", %" #r ", %8\n"
#define I_MAX64(r) "v_max_f64 %" #r ", |%" #r "|, |%8|\n"
#define I_CMP64(r) "v_cmp_lt_f64 vcc, %" #r ", %8\n"
#define I_RSQ64(r) "v_rsq_f64 %" #r ", %" #r "\n"
#define I_RCP64(r) "v_rcp_f64 %" #r ", %" #r "\n"
#define I_FRACT64(r) "v_fract_f64 %" #r ", %" #r "\n"
#define I_RNDNE64(r) "v_rndne_f64 %" #r ", %" #r "\n"
KERNEL64(fma_f64, I_FMA64)
KERNEL64(add_f64, I_ADD64)
KERNEL64(mul_f64, I_MUL64)
KERNEL64(max_f64, I_MAX64)
KERNEL64(cmp_f64, I_CMP64)
KERNEL64(rsq_f64, I_RSQ64)
KERNEL64(rcp_f64, I_RCP64)
KERNEL64(fract_f64, I_FRACT64)
KERNEL64(rndne_f64, I_RNDNE64)

/* f32 / int */
#define I_FMA32(r) "v_fma_f32 %" #r ", %" #r ", %8, %9\n"
#define I_ADD32(r) "v_add_f32 %" #r ", %" #r ", %8\n"
#define I_MUL32(r) "v_mul_f32 %" #r ", %" #r ", %8\n"
#define I_MAX332(r) "v_max3_f32 %" #r ", |%" #r "|, |%8|, |%9|\n"
#define I_CMP32(r) "v_cmp_lt_f32 vcc, %" #r ", %8\n"
#define I_RSQ32(r) "v_rsq_f32 %" #r ", %" #r "\n"
#define I_CVTF32I32(r) "v_cvt_f32_i32 %" #r ", %" #r "\n"
#define I_SUBU32(r) "v_sub_u32 %" #r ", %" #r ", %8\n"
#define I_CMPU32(r) "v_cmp_lt_u32 vcc, %" #r ", %8\n"
#define I_CNDMASK(r) "v_cndmask_b32 %" #r ", %" #r ", %8, vcc\n"
#define I_CNDMASK_S(r) "v_cndmask_b32_e64 %" #r ", %" #r ", %8, s[20:21]\n"
#define I_OR32(r) "v_or_b32 %" #r ", %" #r ", %8\n"
#define I_LSHLOR(r) "v_lshl_or_b32 %" #r ", %" #r ", 1, %8\n"
#define I_FFBL(r) "v_ffbl_b32 %" #r ", %" #r "\n"
#define I_MOV(r) "v_mov_b32 %" #r ", %8\n"
#define I_CMP32S(r) "v_cmp_lt_f32_e64 s[20:21], %" #r ", %8\n"
#define I_CND_S(r) "v_cndmask_b32_e64 %" #r ", %" #r ", %8, s[20:21]\n"
#define I_ADDU32(r) "v_add_u32 %" #r ", %" #r ", %8\n"
#define I_AND32(r) "v_and_b32 %" #r ", %" #r ", %8\n"
#define I_LSHL32(r) "v_lshlrev_b32 %" #r ", 1, %" #r "\n"
#define I_BFE(r) "v_bfe_u32 %" #r ", %" #r ", 1, 8\n"
#define I_MAX32(r) "v_max_f32 %" #r ", %" #r ", %8\n"
#define I_MAX32ABS(r) "v_max_f32_e64 %" #r ", |%" #r "|, |%8|\n"
#define I_CMPX32(r) "v_cmp_gt_f32 vcc, %8, %" #r "\n v_add_f32 %" #r ", %" #r ", %8\n"
KERNEL32(cmp_f32_sgpr, I_CMP32S)
KERNEL32(add_u32, I_ADDU32)
KERNEL32(and_b32, I_AND32)
KERNEL32(lshl_b32, I_LSHL32)
KERNEL32(bfe_u32, I_BFE)
KERNEL32(max_f32, I_MAX32)
KERNEL32(max_f32_abs, I_MAX32ABS)
KERNEL32(cmp_add_f32_pair, I_CMPX32)
KERNEL32(fma_f32, I_FMA32)
KERNEL32(add_f32, I_ADD32)
KERNEL32(mul_f32, I_MUL32)
KERNEL32(max3_f32, I_MAX332)
KERNEL32(cmp_f32, I_CMP32)
KERNEL32(rsq_f32, I_RSQ32)
KERNEL32(cvt_f32_i32, I_CVTF32I32)
KERNEL32(sub_u32, I_SUBU32)
KERNEL32(cmp_u32, I_CMPU32)
KERNEL32(cndmask, I_CNDMASK)
KERNEL32(cndmask_sgpr, I_CNDMASK_S)
KERNEL32(or_b32, I_OR32)
KERNEL32(lshl_or, I_LSHLOR)
KERNEL32(ffbl, I_FFBL)
KERNEL32(mov_b32, I_MOV)

/* conversions and packed forms need their own register shapes */
__global__ __launch_bounds__(256) void k_cvt_f32_f64(unsigned long long *out, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float f0 = 0, f1 = 0, f2 = 0, f3 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for(int it = 0; it < ITER; it++) {
        asm volatile(REP8("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n") : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if((threadIdx.x & 63) == 0)
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if(f0 + f1 + f2 + f3 == 1.2345f)
        out[0] = 0;
}
__global__ __launch_bounds__(256) void k_cvt_f64_f32(unsigned long long *out, double seed)
{
    double a0 = 0, a1 = 0;
    float f0 = (float) seed + threadIdx.x, f1 = f0 + 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for(int it = 0; it < ITER; it++) {
        asm volatile(REP8("v_cvt_f64_f32 %0, %2\n v_cvt_f64_f32 %1, %3\n") : "=v"(a0), "=v"(a1) : "v"(f0), "v"(f1));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if((threadIdx.x & 63) == 0)
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if(a0 + a1 == 1.2345)
        out[0] = 0;
}
__global__ __launch_bounds__(256) void k_cvt_i32_f64(unsigned long long *out, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1;
    int f0 = 0, f1 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for(int it = 0; it < ITER; it++) {
        asm volatile(REP8("v_cvt_i32_f64 %0, %2\n v_cvt_i32_f64 %1, %3\n") : "=v"(f0), "=v"(f1) : "v"(a0), "v"(a1));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if((threadIdx.x & 63) == 0)
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if(f0 + f1 == 12345)
        out[0] = 0;
}
__global__ __launch_bounds__(256) void k_pk_fma_f32(unsigned long long *out, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = seed * 0.5 + 1.0, c = seed * 0.25 + 0.125;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for(int it = 0; it < ITER; it++) {
#define I_PK(r) "v_pk_fma_f32 %" #r ", %" #r ", %8, %9\n"
        asm volatile(I_PK(0) I_PK(1) I_PK(2) I_PK(3) I_PK(4) I_PK(5) I_PK(6) I_PK(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c));
        asm volatile(I_PK(0) I_PK(1) I_PK(2) I_PK(3) I_PK(4) I_PK(5) I_PK(6) I_PK(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if((threadIdx.x & 63) == 0)
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345)
        out[0] = 0;
}

/* LDS reads: 16 B per lane; mode 0: every lane its own slot (stride 32 B), mode 1: all lanes one address (broadcast),
 * mode 2: pseudo-random 32-B slots out of 64 */
template <int MODE>
__global__ __launch_bounds__(256) void k_ds_read_b128(unsigned long long *out, double seed)
{
    __shared__ double4 buf[4][128];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    buf[w][lane] = make_double4(seed, 1, 2, 3);
    buf[w][lane + 64] = make_double4(seed, 1, 2, 3);
    __syncthreads();
    int slot = MODE == 0 ? lane : (MODE == 1 ? 5 : ((lane * 37 + 11) & 63));
    const __attribute__((address_space(3))) double4 *p = (const __attribute__((address_space(3))) double4 *) &buf[w][slot];
    typedef float f4 __attribute__((ext_vector_type(4)));
    double s = 0;
    const unsigned addr = (unsigned) (size_t) p; /* LDS byte address: low 32 bits of the generic-to-local pointer */
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for(int it = 0; it < ITER; it++) {
        f4 q0, q1, q2, q3;
        asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:2048\n ds_read_b128 %2, %4 offset:16\n ds_read_b128 %3, %4 offset:2064\n"
                     "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:2048\n ds_read_b128 %2, %4 offset:16\n ds_read_b128 %3, %4 offset:2064\n"
                     "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:2048\n ds_read_b128 %2, %4 offset:16\n ds_read_b128 %3, %4 offset:2064\n"
                     "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:2048\n ds_read_b128 %2, %4 offset:16\n ds_read_b128 %3, %4 offset:2064\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(addr));
        s += q0.x + q1.x + q2.x + q3.x;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if((threadIdx.x & 63) == 0)
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if(s == 1.2345)
        out[0] = 0;
}

typedef void (*kern_t)(unsigned long long *, double);
struct Entry { const char *name; kern_t k; };
#define E(n) {#n, k_##n}

int main(int argc, char **argv)
{
    Entry tab[] = {E(fma_f64), E(add_f64), E(mul_f64), E(max_f64), E(cmp_f64), E(rsq_f64), E(rcp_f64), E(fract_f64), E(rndne_f64),
                   E(cvt_f32_f64), E(cvt_f64_f32), E(cvt_i32_f64), E(fma_f32), E(cmp_f32_sgpr), E(add_u32), E(and_b32), E(lshl_b32), E(bfe_u32), E(max_f32), E(max_f32_abs), E(cmp_add_f32_pair), E(add_f32), E(mul_f32), E(max3_f32), E(cmp_f32),
                   E(rsq_f32), E(cvt_f32_i32), E(pk_fma_f32), E(sub_u32), E(cmp_u32), E(cndmask), E(cndmask_sgpr), E(or_b32), E(lshl_or), E(ffbl),
                   E(mov_b32), {"ds_read_b128_own", k_ds_read_b128<0>}, {"ds_read_b128_bcast", k_ds_read_b128<1>},
                   {"ds_read_b128_rand", k_ds_read_b128<2>}};
    int ncu = 256;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, ncu);
    unsigned long long *d_out, *h_out;
    const int maxblocks = ncu * 8;
    hipMalloc(&d_out, sizeof(unsigned long long) * maxblocks * 4);
    h_out = (unsigned long long *) malloc(sizeof(unsigned long long) * maxblocks * 4);
    printf("%-20s %8s %8s %8s %8s   (shader cycles per wave-instruction per SIMD; 16 instr x %d iterations)\n", "instruction", "w=1", "w=2", "w=4", "w=8", ITER);
    for(auto &e : tab) {
        printf("%-20s", e.name);
        for(int w = 1; w <= 8; w *= 2) {
            const int blocks = ncu * w;
            hipMemset(d_out, 0, sizeof(unsigned long long) * blocks * 4);
            e.k<<<blocks, 256>>>(d_out, 1.5); /* warm */
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            e.k<<<blocks, 256>>>(d_out, 1.5);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h_out, d_out, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
            double sum = 0;
            for(int i = 0; i < blocks * 4; i++)
                sum += (double) h_out[i];
            const double cyc_per_wave = sum / (blocks * 4.0);
            /* per SIMD: w waves each issue 16*ITER instructions in cyc_per_wave cycles */
            const double per = cyc_per_wave / (16.0 * ITER * w);
            printf(" %8.2f", per);
            if(w == 8)
                printf("   [w=8: %.3f ms, %.0f ticks per wave => tick rate %.2f GHz if the waves ran the whole launch]", ms, cyc_per_wave, cyc_per_wave / (ms * 1e6));
        }
        printf("\n");
    }
    return 0;
}
