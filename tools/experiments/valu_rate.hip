// Issue rates of the vector instructions the operand conversions are made of: alone, next to a wave that runs MFMAs on
// the same SIMD, and with both waves of the SIMD running the same instruction stream.  One workgroup of 512 threads on one CU: waves 0 .. 3 (one per SIMD) run 64 x 32 independent copies of one
// instruction; waves 4 .. 7 either idle or run back-to-back v_mfma_f32_16x16x32_f16.  Clocks per instruction from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o build/valu_rate && build/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define REP32(X) X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X

template <int OP>
__device__ __forceinline__ void body(unsigned &a, unsigned &b, unsigned &c, unsigned &d, unsigned long long &q)
{
    f4 r4;
    // every instruction reads loop-invariant sources and writes a register nobody reads: no dependences, pure issue rate
    if (OP == 0) { REP32(asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(a) : "v"(b), "v"(c), "v"(d));) }
    if (OP == 1) { REP32(asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));) }
    if (OP == 2) { REP32(asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(q) : "v"(q));) }
    if (OP == 3) { REP32(asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a) : "v"(b));) }
    if (OP == 4) { REP32(asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));) }
    if (OP == 5) { REP32(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q) : "v"(b), "v"(c) : "vcc");) }
    if (OP == 6) { REP32(asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));) }
    if (OP == 7) { REP32(asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(b), "v"(c), "v"(d));) }
    if (OP == 8) { REP32(asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(a) : "v"(b));) }
    if (OP == 9) { REP32(asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(a) : "v"(b), "v"(c), "v"(d));) }
    if (OP == 10) { REP32(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c));) }
    if (OP == 11) { REP32(asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(q));) }
    if (OP == 12) { REP32(asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(q) : "v"(q));) }
    if (OP == 13) { REP32(asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a) : "v"(b));) }
    if (OP == 14) { REP32(asm volatile("s_add_u32 s20, s21, s22" ::: "s20", "scc");) }
    if (OP == 15) { REP32(asm volatile("v_fma_f32 %0, %1, %2, %3\n\ts_add_u32 s20, s21, s22" : "=v"(a) : "v"(b), "v"(c), "v"(d) : "s20", "scc");) } // 64 instructions
    if (OP == 16) { REP32(asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));) }
    if (OP == 17) { REP32(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(d));) } // a dependent chain
    if (OP == 19) { REP32(asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(a) : "v"(b), "v"(c));) }
    if (OP == 20) { REP32(asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(b), "v"(c) : "vcc");) }
    if (OP == 21) { REP32(asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c) : "vcc");) }
    if (OP == 22) { REP32(asm volatile("v_cmp_lt_f32_e64 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(a) : "v"(b), "v"(c) : "s20", "s21");) }
    if (OP == 23) { REP32(asm volatile("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));) }
    if (OP == 24) { REP32(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a) : "v"(b));) }
    if (OP == 25) { REP32(asm volatile("v_exp_f32 %0, %1" : "=v"(a) : "v"(b));) }
    if (OP == 26) { REP32(asm volatile("v_rcp_f32 %0, %1" : "=v"(a) : "v"(b));) }
    if (OP == 27) { REP32(asm volatile("ds_read_b128 %0, %1" : "=v"(*(f4 *)&r4) : "v"(b & 0xff0));) }
    if (OP == 18) { REP32(asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(a));) }
}

template <int OP>
__global__ __launch_bounds__(512, 1) void k(unsigned long long *out, int with_mfma, int iters)
{
    const int w = threadIdx.x >> 6;
    unsigned a = threadIdx.x, b = 0x3c003c00u + threadIdx.x, c = 0x40000000u, d = 0x3c00u;
    unsigned long long q = threadIdx.x;
    f4 acc[8];
    for (int i = 0; i < 8; ++i)
        acc[i] = (f4){0, 0, 0, 0};
    h8 x, y;
    for (int i = 0; i < 8; ++i) {
        x[i] = (_Float16)(float)(threadIdx.x & 3);
        y[i] = (_Float16)1.0f;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (w < 4 || with_mfma == 2) {
        for (int it = 0; it < iters; ++it)
            body<OP>(a, b, c, d, q);
    } else if (with_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0)
        out[w] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 8; ++i)
        s += acc[i][0];
    if (s == 12345.0f || a == 0xdeadbeefu || q == 77)
        out[8] = a + (unsigned)q; // keep everything alive
}

template <int OP>
void run(const char *name, unsigned long long *d)
{
    const int iters = 2000;
    unsigned long long h[8];
    double v[3], m = 0;
    for (int mf = 0; mf < 3; ++mf) {
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(512), 0, 0, d, mf, iters);
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(512), 0, 0, d, mf, iters);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        v[mf] = (double)h[0] / (iters * 32.0);
        if (mf == 1)
            m = (double)h[4] / (iters * 32.0);
    }
    printf("%-22s clocks / instruction: alone %6.2f   next to MFMAs %6.2f   (an MFMA of that run: %6.2f)   two waves per SIMD running it: %6.2f each\n", name, v[0], v[1], m, v[2]);
}

int main()
{
    unsigned long long *d;
    hipMalloc(&d, 16 * sizeof(unsigned long long));
    run<7>("v_fma_f32", d);
    run<0>("v_fma_mixlo_f16", d);
    run<9>("v_fma_mixhi_f16", d);
    run<1>("v_cvt_pk_f16_f32", d);
    run<8>("v_cvt_f16_f32", d);
    run<3>("v_cvt_f32_f16", d);
    run<13>("v_cvt_f32_f16_sdwa", d);
    run<2>("v_pk_mul_f32", d);
    run<12>("v_pk_fma_f32", d);
    run<10>("v_cndmask_b32", d);
    run<4>("v_mul_lo_u32", d);
    run<6>("v_mul_u32_u24", d);
    run<5>("v_mad_u64_u32", d);
    run<11>("v_lshl_add_u64", d);
    run<16>("v_mov_b32", d);
    run<17>("v_fma_f32 (dependent)", d);
    run<18>("v_accvgpr_read_b32", d);
    run<10>("v_cndmask_b32 (vcc)", d);
    run<19>("v_cndmask_b32_e64 (sgpr pair)", d);
    run<20>("v_cmp_lt_f32 -> vcc", d);
    run<21>("v_cmp + v_cndmask (vcc) per pair / 2", d);
    run<22>("v_cmp + v_cndmask (sgpr) per pair / 2", d);
    run<23>("v_max_f32", d);
    run<24>("v_mov_b32_dpp", d);
    run<25>("v_exp_f32", d);
    run<26>("v_rcp_f32", d);
    run<27>("ds_read_b128", d);
    run<14>("s_add_u32", d);
    run<15>("v_fma_f32 + s_add_u32 (per pair / 2)", d);
    return 0;
}
