// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of THIS repo's kernels (the guide calibrates it for wide
// coalesced streaming reads only: "exactly half of the bytes"; "other access widths are uncalibrated").  Every kernel reads the
// same N bytes of a buffer larger than the 256 MiB Infinity Cache exactly once, in a different shape:
//   k16   16 bytes per lane, 16-byte aligned (the sweeps' tile loads)
//   k4    4 bytes per lane, contiguous (global_load_dword: plain one-float-per-lane rows)
//   kb4   4 bytes per lane through a raw buffer descriptor with per-lane byte offsets (the gathers of the source-fused level 0)
//   kb1   1 byte per lane through a raw buffer descriptor (unsigned char frames)
//   ko16  16 bytes per lane at 4-byte alignment, lanes 8 bytes apart (k_collapse4's windows of the coarser row: neighbours overlap by half)
// usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib.bin ; the tool prints N.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__global__ void k16(const float* __restrict__ in, size_t n4, float* out) {
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f4 v = reinterpret_cast<const f4*>(in)[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1.2345f) out[0] = acc;
}
__global__ void k4(const float* __restrict__ in, size_t n, float* out) {
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
    if (acc == 1.2345f) out[0] = acc;
}
// the buffer kernels walk 1 GiB windows (a descriptor addresses 4 GiB at most)
__global__ void kb4(const float* __restrict__ in, size_t n, float* out) {
    float acc = 0;
    const __amdgpu_buffer_rsrc_t r = rsrc(in, (unsigned)(n * 4));
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (unsigned)(i * 4), 0, 0));
    if (acc == 1.2345f) out[0] = acc;
}
__global__ void kb1(const uint8_t* __restrict__ in, size_t n, float* out) {
    float acc = 0;
    const __amdgpu_buffer_rsrc_t r = rsrc(in, (unsigned)n);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += (float)__builtin_amdgcn_raw_buffer_load_b8(r, (unsigned)i, 0, 0);
    if (acc == 1.2345f) out[0] = acc;
}
__global__ void ko16(const float* __restrict__ in, size_t n, float* out) {  // lane i of a wavefront reads floats 2i .. 2i+3 of its 128-float segment (+2)
    float acc = 0;
    const size_t nseg = n / 128;
    for (size_t s = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) / 64; s < nseg - 1; s += (size_t)gridDim.x * blockDim.x / 64) {
        const f4 v = *reinterpret_cast<const f4u*>(in + s * 128 + 2 * (threadIdx.x & 63));
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1.2345f) out[0] = acc;
}
int main() {
    const size_t N = (size_t)1 << 30;  // bytes per kernel
    float *buf, *out;
    hipMalloc(&buf, N);
    hipMalloc(&out, 64);
    hipMemset(buf, 0, N);
    hipDeviceSynchronize();
    const int G = 8192, B = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k16, dim3(G), dim3(B), 0, 0, buf, N / 16, out);
        hipLaunchKernelGGL(k4, dim3(G), dim3(B), 0, 0, buf, N / 4, out);
        hipLaunchKernelGGL(kb4, dim3(G), dim3(B), 0, 0, buf, N / 4, out);
        hipLaunchKernelGGL(kb1, dim3(G), dim3(B), 0, 0, (const uint8_t*)buf, N, out);
        hipLaunchKernelGGL(ko16, dim3(G), dim3(B), 0, 0, buf, N / 4, out);
        hipDeviceSynchronize();
    }
    printf("every kernel reads %zu bytes = %zu KiB once per launch\n", N, N / 1024);
    return 0;
}
