// One wavefront per CU running k_vv_y_bwd_dec7's chain loop in isolation: what does a row cost when nothing else is on the SIMD?
//   A  8 rows per iteration from registers only (no LDS): cvt, x*sum, 3 x (mul, add), cvt per row
//   B  as A, rows read from / written to an LDS slot by column (ds_read_b32 / ds_write_b32, stride 128 floats), next chunk's reads
//      issued before this chunk's arithmetic
//   C  as B plus what the kernel does per chunk besides: a relaxed and a release LDS atomic, the compare-and-branch on the loader's
//      counter (a second wavefront bumps it)
//   D  as B with 16 rows per iteration
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off chain_rate2.hip -o chain_rate2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
struct K { double sum, f1, f2, f3; };
__device__ __forceinline__ float step(float x, const K& k, double& v1, double& v2, double& v3) {
    double v0 = (double)x;
    v0 *= k.sum;
    v0 += v1 * k.f1;
    v0 += v2 * k.f2;
    v0 += v3 * k.f3;
    v3 = v2; v2 = v1; v1 = v0;
    return (float)v0;
}
__global__ __launch_bounds__(64) void k_a(float* out, int chunks, K k) {
    double v1 = 1 + threadIdx.x, v2 = 1, v3 = 1;
    float xv[8];
    for (int u = 0; u < 8; ++u) xv[u] = 1.f + u;
    for (int j = 0; j < chunks; ++j) {
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = step(xv[u], k, v1, v2, v3);
    }
    float s = 0;
    for (int u = 0; u < 8; ++u) s += xv[u];
    out[blockIdx.x * 64 + threadIdx.x] = s + (float)v1;
}
template <int ROWS>
__global__ __launch_bounds__(64) void k_b(float* out, int chunks, K k) {
    __shared__ float ring[8][ROWS][128];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8 * ROWS * 128; i += 64) (&ring[0][0][0])[i] = 1.f + (i % 5);
    __syncthreads();
    double v1 = 1 + lane, v2 = 1, v3 = 1;
    float xv[ROWS], nx[ROWS];
#pragma unroll
    for (int u = 0; u < ROWS; ++u) xv[u] = ring[0][u][lane];
    for (int j = 0; j < chunks; ++j) {
        const float* c1 = &ring[(j + 1) & 7][0][lane];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) nx[u] = c1[u * 128];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) xv[u] = step(xv[u], k, v1, v2, v3);
        float* col = &ring[j & 7][0][lane];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) col[u * 128] = xv[u];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) xv[u] = nx[u];
    }
    out[blockIdx.x * 64 + lane] = (float)v1 + ring[3][1][lane];
}
__global__ __launch_bounds__(128) void k_c(float* out, int chunks, K k) {
    __shared__ float ring[8][8][128];
    __shared__ int cnt_loaded, cnt_chain;
    typedef __attribute__((address_space(3))) int lds_int;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8 * 8 * 128; i += 128) (&ring[0][0][0])[i] = 1.f + (i % 5);
    if (threadIdx.x == 0) cnt_loaded = 0, cnt_chain = 0;
    __syncthreads();
    if (wave == 1) {  // the "loader": stays a few chunks ahead of the chain
        for (int j = 0; j < chunks; ++j) {
            while (__hip_atomic_load((lds_int*)&cnt_chain, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < j - 5) __builtin_amdgcn_s_sleep(2);
            __hip_atomic_store((lds_int*)&cnt_loaded, j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    double v1 = 1 + lane, v2 = 1, v3 = 1;
    float xv[8], nx[8];
    int loaded = 0;
    while (loaded < 1) loaded = __hip_atomic_load((lds_int*)&cnt_loaded, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = ring[0][u][lane], nx[u] = 0.f;
    for (int j = 0; j < chunks; ++j) {
        const int seen = __hip_atomic_load((lds_int*)&cnt_loaded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const bool early = j + 1 < chunks && loaded >= j + 2;
        asm volatile("" ::: "memory");
        if (early) {
            const float* c1 = &ring[(j + 1) & 7][0][lane];
#pragma unroll
            for (int u = 0; u < 8; ++u) nx[u] = c1[u * 128];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = step(xv[u], k, v1, v2, v3);
        if (j >= 1) __hip_atomic_store((lds_int*)&cnt_chain, j, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int u = 4; u < 8; ++u) xv[u] = step(xv[u], k, v1, v2, v3);
        float* col = &ring[j & 7][0][lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) col[u * 128] = xv[u];
        loaded = max(loaded, seen);
        if (j + 1 < chunks && !early) {
            while (loaded < j + 2) {
                loaded = __hip_atomic_load((lds_int*)&cnt_loaded, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (loaded < j + 2) __builtin_amdgcn_s_sleep(1);
            }
            const float* c1 = &ring[(j + 1) & 7][0][lane];
#pragma unroll
            for (int u = 0; u < 8; ++u) nx[u] = c1[u * 128];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = nx[u];
    }
    __hip_atomic_store((lds_int*)&cnt_chain, chunks, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    out[blockIdx.x * 64 + lane] = (float)v1 + ring[3][1][lane];
}
template <typename F>
float time_ms(F&& launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* d;
    (void)hipMalloc(&d, 1024 * 64 * sizeof(float));
    const K k{0.9, 1.2e-3, -3.1e-4, 7.7e-5};
    for (int wgs : {256, 32}) {
        for (int chunks : {256, 20000}) {
            const double rows = 8.0 * chunks;
            const float a = time_ms([&] { hipLaunchKernelGGL(k_a, dim3(wgs), dim3(64), 0, 0, d, chunks, k); });
            const float b = time_ms([&] { hipLaunchKernelGGL(k_b<8>, dim3(wgs), dim3(64), 0, 0, d, chunks, k); });
            const float c = time_ms([&] { hipLaunchKernelGGL(k_c, dim3(wgs), dim3(128), 0, 0, d, chunks, k); });
            const float e = time_ms([&] { hipLaunchKernelGGL(k_b<16>, dim3(wgs), dim3(64), 0, 0, d, chunks / 2, k); });
            printf("%3d workgroups, %5d chunks of 8 rows: A registers only %.1f ns/row | B LDS columns %.1f | C + counters and a loader wavefront %.1f | D 16-row chunks %.1f   (launch %.1f us for A)\n",
                   wgs, chunks, a * 1e6 / rows, b * 1e6 / rows, c * 1e6 / rows, e * 1e6 / rows, a * 1e3);
        }
    }
    return 0;
}
