// Issue cost of vector instructions on gfx950, as this repo's kernels use them: N independent instructions of one kind per
// loop trip (eight accumulators, so that no instruction waits for the one before it), timed with s_memtime on wavefront 0 of
// each workgroup, with 1, 2 and 4 wavefronts per SIMD (workgroups of 256, 512 and 1024 work-items on one CU each).
// build: hipcc -O2 --offload-arch=gfx950 valu_rate.hip -o valu_rate.bin ; run: ./valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int KIND>
__global__ void k_rate(unsigned long long* out, int trips, double seed_d, float seed_f, unsigned seed_u) {
    double d[8];
    float f[8];
    unsigned u[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
    for (int i = 0; i < 8; ++i) d[i] = seed_d + i + threadIdx.x, f[i] = seed_f + i + threadIdx.x, u[i] = seed_u + i * 77 + threadIdx.x, p[i] = f2{f[i], f[i] + 1};
    const unsigned long long mask64 = 0x5555555555555555ull ^ (unsigned long long)seed_u;
    unsigned long long m8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double cd = seed_d * 0.999;
    const float cf = seed_f * 0.999f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < trips; ++t) {
#define ADD_F32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(cf));
#define FMA_F32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(cf));
#define MUL_F64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
#define ADD_F64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
#define FMA_F64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(cd));
#define CVT_F64_F32(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
#define CVT_F32_F64(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
#define PK_MUL_F32(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
#define PK_FMA_F32(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
#define MUL_LO_U32(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(seed_u));
#define MUL_HI_U32(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(seed_u));
#define MAD_U24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(seed_u));
#define CVT_UB(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[i]) : "v"(u[i]));
#define RCP_F32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
#define CVT_I32_F32(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(f[i]));
#define PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(seed_u));
#define CNDMASK(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(seed_u), "s"(mask64));
#define CMPCND(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(u[i]) : "v"(f[i]), "v"(cf), "v"(seed_u) : "vcc");
#define CMP64(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m8[i]) : "v"(f[i]), "v"(cf));
#define MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(cf));
#define MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(cf));
#define FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(f[i]));
#define FMA_F64_DEP(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[0]) : "v"(cd));
#define ADD_F64_DEP(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[0]) : "v"(cd));
#define ADD_F32_DEP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[0]) : "v"(cf));
        if (KIND == 0) { REP8(ADD_F32) REP8(ADD_F32) }
        if (KIND == 1) { REP8(FMA_F32) REP8(FMA_F32) }
        if (KIND == 2) { REP8(MUL_F64) REP8(MUL_F64) }
        if (KIND == 3) { REP8(ADD_F64) REP8(ADD_F64) }
        if (KIND == 4) { REP8(FMA_F64) REP8(FMA_F64) }
        if (KIND == 5) { REP8(CVT_F64_F32) REP8(CVT_F64_F32) }
        if (KIND == 6) { REP8(CVT_F32_F64) REP8(CVT_F32_F64) }
        if (KIND == 7) { REP8(PK_MUL_F32) REP8(PK_MUL_F32) }
        if (KIND == 8) { REP8(PK_FMA_F32) REP8(PK_FMA_F32) }
        if (KIND == 9) { REP8(MUL_LO_U32) REP8(MUL_LO_U32) }
        if (KIND == 10) { REP8(MUL_HI_U32) REP8(MUL_HI_U32) }
        if (KIND == 11) { REP8(MAD_U24) REP8(MAD_U24) }
        if (KIND == 12) { REP8(CVT_UB) REP8(CVT_UB) }
        if (KIND == 13) { REP8(RCP_F32) REP8(RCP_F32) }
        if (KIND == 14) { REP8(CVT_I32_F32) REP8(CVT_I32_F32) }
        if (KIND == 15) { REP8(PERM) REP8(PERM) }
        if (KIND == 16) { REP8(CNDMASK) REP8(CNDMASK) }
        if (KIND == 17) { REP8(FLOOR) REP8(FLOOR) }
        if (KIND == 18) { REP8(FMA_F64_DEP) REP8(FMA_F64_DEP) }
        if (KIND == 19) { REP8(ADD_F64_DEP) REP8(ADD_F64_DEP) }
        if (KIND == 20) { REP8(ADD_F32_DEP) REP8(ADD_F32_DEP) }
        if (KIND == 21) { REP8(CMPCND) REP8(CMPCND) }
        if (KIND == 22) { REP8(CMP64) REP8(CMP64) }
        if (KIND == 23) { REP8(MED3) REP8(MED3) }
        if (KIND == 24) { REP8(MAXF) REP8(MAXF) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double acc = 0;
    for (int i = 0; i < 8; ++i) acc += d[i] + f[i] + u[i] + p[i].x + p[i].y + (double)m8[i];
    if (acc == 12345.678) out[1023] = 1;  // keep the values alive
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
static const char* NAMES[] = {"v_add_f32", "v_fma_f32", "v_mul_f64", "v_add_f64", "v_fma_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_pk_mul_f32",
                              "v_pk_fma_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_cvt_f32_ubyte1", "v_rcp_f32", "v_cvt_i32_f32",
                              "v_perm_b32", "v_cndmask_b32_e64 (SGPR mask)", "v_floor_f32", "v_fma_f64 (dependent chain)", "v_add_f64 (dependent chain)",
                              "v_add_f32 (dependent chain)", "v_cmp_lt_f32 + s_nop 1 + v_cndmask (per pair)", "v_cmp_lt_f32_e64 -> SGPR pair", "v_med3_f32", "v_max_f32"};
template <int KIND>
void run(unsigned long long* d_out) {
    const int trips = 20000;
    printf("%-30s", NAMES[KIND]);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int waves_per_simd : {1, 2, 4}) {
        const int block = 256 * waves_per_simd;
        hipLaunchKernelGGL(k_rate<KIND>, dim3(8), dim3(block), 0, 0, d_out, trips, 1.000001, 1.0001f, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(8), dim3(block), 0, 0, d_out, trips, 1.000001, 1.0001f, 12345u);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[8];
        hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
        double t = 0;
        for (int i = 0; i < 8; ++i) t += (double)h[i];
        t /= 8;
        // s_memtime counts at 100 MHz on gfx9: report both raw ticks and ticks per instruction issued by the SIMD
        // wall time per instruction issued by a SIMD, in ns (events; launch overhead ~5 us included) and s_memtime ticks per wave-instruction
        printf("  %dw/SIMD: %6.3f ns/instr/SIMD (%5.2f ticks/instr/wave, %.0f ticks/us)", waves_per_simd, ms * 1e6 / (trips * 16.0 * waves_per_simd),
               t / (trips * 16.0), t / (ms * 1e3));
    }
    printf("\n");
}
int main() {
    unsigned long long* d_out;
    hipMalloc(&d_out, 1024 * sizeof(unsigned long long));
    run<0>(d_out); run<1>(d_out); run<2>(d_out); run<3>(d_out); run<4>(d_out); run<5>(d_out); run<6>(d_out); run<7>(d_out); run<8>(d_out);
    run<9>(d_out); run<10>(d_out); run<11>(d_out); run<12>(d_out); run<13>(d_out); run<14>(d_out); run<15>(d_out); run<16>(d_out); run<17>(d_out);
    run<18>(d_out); run<19>(d_out); run<20>(d_out); run<21>(d_out); run<22>(d_out); run<23>(d_out); run<24>(d_out);
    return 0;
}
