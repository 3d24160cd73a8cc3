// micro-benchmark: streaming read / write / copy bandwidth on MI355X by access width (4, 8, 16 B per lane)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <typename T> __global__ __launch_bounds__(256) void k_read(const T* __restrict__ p, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0;
    for (; i < n; i += stride) { T v = p[i]; const float* f = reinterpret_cast<const float*>(&v);
        for (int k = 0; k < (int)(sizeof(T)/4); ++k) acc += f[k]; }
    if (acc == 123.456f) *sink = acc;
}
template <typename T> __global__ __launch_bounds__(256) void k_write(T* __restrict__ p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    T v; float* f = reinterpret_cast<float*>(&v); for (int k = 0; k < (int)(sizeof(T)/4); ++k) f[k] = 1.f;
    for (; i < n; i += stride) p[i] = v;
}
template <typename T> __global__ __launch_bounds__(256) void k_copy(const T* __restrict__ a, T* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) b[i] = a[i];
}
// 7 read streams + 3 write streams, one element per thread per stream (the collapse pattern), no grid stride
template <typename T> __global__ __launch_bounds__(256) void k_7r3w(const T* __restrict__ a, T* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T s0 = a[i], s1 = a[i + n], s2 = a[i + 2*n], s3 = a[i + 3*n], s4 = a[i + 4*n], s5 = a[i + 5*n], s6 = a[i + 6*n];
    b[i] = s0 + s3 * s6; b[i + n] = s1 + s4 * s6; b[i + 2*n] = s2 + s5 * s6;
}
template <typename F> float timeit(F f, int reps = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const size_t bytes = (size_t)1 << 30;  // 1 GiB per buffer
    float *A, *B, *sink; hipMalloc(&A, bytes * 2); hipMalloc(&B, bytes); hipMalloc(&sink, 4);
    hipMemset(A, 0, bytes * 2); hipMemset(B, 0, bytes);
    for (int grid : {2048, 8192, 65536}) {
        printf("grid %d\n", grid);
        #define RUN(T, name) { size_t n = bytes / sizeof(T); \
            float r = timeit([&]{ k_read<T><<<grid,256>>>((const T*)A, n, sink); }); \
            float w = timeit([&]{ k_write<T><<<grid,256>>>((T*)B, n); }); \
            float c = timeit([&]{ k_copy<T><<<grid,256>>>((const T*)A, (T*)B, n); }); \
            printf("  %-6s read %7.1f GB/s  write %7.1f GB/s  copy(R+W) %7.1f GB/s\n", name, bytes/r/1e6, bytes/w/1e6, 2.0*bytes/c/1e6); }
        RUN(float, "4B") RUN(f2, "8B") RUN(f4, "16B")
    }
    {   // 7r3w: planes of 100 MB
        size_t plane = 100u << 20;
        #define RUN2(T, name) { size_t n = plane / sizeof(T); int g = (int)((n + 255) / 256); \
            float t = timeit([&]{ k_7r3w<T><<<g,256>>>((const T*)A, (T*)B, n); }); \
            printf("  7r3w %-4s %7.1f GB/s (R+W)\n", name, 10.0*plane/t/1e6); }
        RUN2(float, "4B") RUN2(f2, "8B") RUN2(f4, "16B")
    }
    return 0;
}
