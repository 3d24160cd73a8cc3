// What does ONE wavefront that has a SIMD to itself pay per step of the Van Vliet recurrence, depending on where the converts live?
//   A  floats in LDS (today's k_vv_x_fwd): ds_read_b128 of 4 floats, per sample cvt f32->f64, 3 mul + 3 add, cvt f64->f32, ds_write_b128
//   B  doubles in LDS, converts done by ANOTHER wavefront: ds_read_b128 of 2 doubles, per sample 3 mul + 3 add, ds_write_b128 of 2 doubles
//   C  as B, with a helper wavefront of the same workgroup (another SIMD of the CU) converting tiles through LDS all the time
// One workgroup per CU (256), `tiles` tiles of 64 samples per lane; time from HIP events -> ns per sample of the chain.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off chain_rate.hip -o chain_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#pragma clang fp contract(off)
constexpr int TP = 68, DP = 66;  // padded LDS rows (floats / doubles)

__global__ __launch_bounds__(64) void k_a(float* out, int tiles, double f1, double f2, double f3) {
    __shared__ __attribute__((aligned(16))) float tile[64 * TP];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * TP; i += 64) tile[i] = 1.0f + (float)(i % 7);
    __syncthreads();
    float* row = tile + lane * TP;
    double v1 = 1, v2 = 1, v3 = 1;
    for (int t = 0; t < tiles; ++t) {
        for (int jb = 0; jb < 64; jb += 16) {
            float xs[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f4*>(xs + 4 * q) = *reinterpret_cast<const f4*>(row + jb + 4 * q);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                double v0 = (double)xs[u];
                v0 += v1 * f1;
                v0 += v2 * f2;
                v0 += v3 * f3;
                xs[u] = (float)v0;
                v3 = v2;
                v2 = v1;
                v1 = v0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f4*>(row + jb + 4 * q) = *reinterpret_cast<const f4*>(xs + 4 * q);
        }
    }
    out[blockIdx.x * 64 + lane] = (float)v1 + row[3];
}

template <bool HELPER>
__global__ __launch_bounds__(128) void k_b(float* out, int tiles, double f1, double f2, double f3) {
    __shared__ __attribute__((aligned(16))) double dt[2][64 * DP];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * 64 * DP; i += blockDim.x) dt[0][i] = 1.0 + (double)(i % 7);
    __syncthreads();
    if (wave == 0) {
        double v1 = 1, v2 = 1, v3 = 1;
        for (int t = 0; t < tiles; ++t) {
            double* row = dt[t & 1] + lane * DP;
            d2 cur[4], nxt[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = *reinterpret_cast<const d2*>(row + 2 * q);
            for (int jb = 0; jb < 64; jb += 8) {
                const int jn = jb + 8 < 64 ? jb + 8 : jb;  // the next group's reads go out before this group's arithmetic
#pragma unroll
                for (int q = 0; q < 4; ++q) nxt[q] = *reinterpret_cast<const d2*>(row + jn + 2 * q);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        double v0 = cur[q][e];
                        v0 += v1 * f1;
                        v0 += v2 * f2;
                        v0 += v3 * f3;
                        cur[q][e] = v0;
                        v3 = v2;
                        v2 = v1;
                        v1 = v0;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<d2*>(row + jb + 2 * q) = cur[q];
#pragma unroll
                for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
            }
            if (HELPER) __syncthreads();
        }
        out[blockIdx.x * 64 + lane] = (float)v1;
    } else if (HELPER) {
        // the helper's share of a tile step: 64 values per lane in (float -> double -> LDS) and 64 out (LDS -> double -> float)
        float acc = 0.f;
        f4 raw[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) raw[i] = f4{1.f + lane, 2.f, 3.f, 4.f + i};
        for (int t = 0; t < tiles; ++t) {
            double* b = dt[(t + 1) & 1] + (lane >> 4) * DP + ((lane & 15) << 2);
#pragma unroll
            for (int i = 0; i < 16; ++i) {  // out: the finished tile
                const d2 a0 = *reinterpret_cast<const d2*>(b + (4 * i) * DP), a1 = *reinterpret_cast<const d2*>(b + (4 * i) * DP + 2);
                acc += (float)a0.x + (float)a0.y + (float)a1.x + (float)a1.y;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {  // in: the next tile
                const f4 v = raw[i];
                *reinterpret_cast<d2*>(b + (4 * i) * DP) = d2{(double)v.x, (double)v.y};
                *reinterpret_cast<d2*>(b + (4 * i) * DP + 2) = d2{(double)v.z, (double)v.w};
                raw[i].x = acc;
            }
            __syncthreads();
        }
        out[(gridDim.x + blockIdx.x) * 64 + lane] = acc;
    }
}

// D: the chain wavefront of k_vv_x_m as it is written there -- x_chain_fwd with run-time sample counts, three rotating tile buffers,
// one workgroup barrier per tile with a second wavefront that only meets the barriers (BAR = false: no second wavefront, no barrier)
__device__ __forceinline__ void x_chain_fwd(float* row, int jmax, bool first_tile, double f1, double f2, double f3, double& v1, double& v2, double& v3) {
    if (first_tile) v1 = v2 = v3 = (double)row[0] / 1.7;
    const int jfull = jmax & ~15;
    for (int jb = 0; jb < jfull; jb += 16) {
        float xs[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f4*>(xs + 4 * q) = *reinterpret_cast<const f4*>(row + jb + 4 * q);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            double v0 = (double)xs[u];
            v0 += v1 * f1;
            v0 += v2 * f2;
            v0 += v3 * f3;
            xs[u] = (float)v0;
            v3 = v2;
            v2 = v1;
            v1 = v0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f4*>(row + jb + 4 * q) = *reinterpret_cast<const f4*>(xs + 4 * q);
    }
    for (int j = jfull; j < jmax; ++j) {
        double v0 = (double)row[j];
        v0 += v1 * f1;
        v0 += v2 * f2;
        v0 += v3 * f3;
        row[j] = (float)v0;
        v3 = v2;
        v2 = v1;
        v1 = v0;
    }
}
template <bool BAR>
__global__ __launch_bounds__(128) void k_d(float* out, int tiles, int w, double f1, double f2, double f3) {
    __shared__ __attribute__((aligned(16))) float tl[3][64 * TP];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 3 * 64 * TP; i += blockDim.x) tl[0][i] = 1.0f + (float)(i % 7);
    __syncthreads();
    if (wave == 0) {
        double v1 = 1, v2 = 1, v3 = 1;
        for (int s = 0; s < tiles; ++s) {
            float* row = tl[s % 3] + lane * TP;
            const int n = min(64, w - s * 64);
            x_chain_fwd(row, n, s == 0, f1, f2, f3, v1, v2, v3);
            if (BAR) __syncthreads();
        }
        out[blockIdx.x * 64 + lane] = (float)v1;
    } else if (BAR) {
        for (int s = 0; s < tiles; ++s) __syncthreads();
    }
}

template <typename F>
float time_ms(F&& launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* d;
    hipMalloc(&d, 2 * 1024 * 64 * sizeof(float));
    const double f1 = 1.2e-3, f2 = -3.1e-4, f3 = 7.7e-5;
    for (int tiles : {35, 512, 2048}) {
        const double n = 64.0 * tiles;
        const float a = time_ms([&] { hipLaunchKernelGGL(k_a, dim3(256), dim3(64), 0, 0, d, tiles, f1, f2, f3); });
        const float b = time_ms([&] { hipLaunchKernelGGL(k_b<false>, dim3(256), dim3(64), 0, 0, d, tiles, f1, f2, f3); });
        const float c = time_ms([&] { hipLaunchKernelGGL(k_b<true>, dim3(256), dim3(128), 0, 0, d, tiles, f1, f2, f3); });
        const float dn = time_ms([&] { hipLaunchKernelGGL(k_d<false>, dim3(256), dim3(64), 0, 0, d, tiles, tiles * 64, f1, f2, f3); });
        const float db = time_ms([&] { hipLaunchKernelGGL(k_d<true>, dim3(256), dim3(128), 0, 0, d, tiles, tiles * 64, f1, f2, f3); });
        const float db2 = time_ms([&] { hipLaunchKernelGGL(k_d<true>, dim3(127), dim3(128), 0, 0, d, tiles, tiles * 64, f1, f2, f3); });
        printf("tiles %5d: A floats in LDS %.2f ns/sample | B doubles in LDS %.2f ns/sample | C doubles + helper wavefront %.2f ns/sample | D k_vv_x_m's chain, no barrier %.2f, "
               "barrier with an idle wavefront %.2f (127 workgroups: %.2f)\n", tiles,
               a * 1e6 / n, b * 1e6 / n, c * 1e6 / n, dn * 1e6 / n, db * 1e6 / n, db2 * 1e6 / n);
    }
    return 0;
}
