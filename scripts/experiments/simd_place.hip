// Where do the wavefronts of a workgroup land, and what does a second double-precision chain on the same SIMD cost?
//  1. k_where: every wavefront of every workgroup records HW_ID (SIMD, CU, SE) and XCC_ID; the host prints, for workgroup sizes of
//     3, 5, 6, 7, 8 wavefronts and 448 workgroups (two per CU on most CUs), the SIMD of each wavefront of the FIRST and the SECOND
//     workgroup of a few CUs, and how many CUs have two "chain" wavefronts (wave 0, or waves 0 and 1) on one SIMD.
//  2. k_chain: workgroups of 8 wavefronts (two per SIMD if placement is round-robin), 256 workgroups; the wavefronts chosen by
//     `mask` run the Van Vliet recurrence (4 dependent fp64 operations per step), the others exit.  ns per step for one chain per
//     CU, two chains on different SIMDs, two chains on the SAME SIMD, four chains on four SIMDs.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off simd_place.hip -o simd_place.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#pragma clang fp contract(off)
__global__ void k_where(unsigned* out, int spin) {
    const int wave = threadIdx.x >> 6;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 16 + wave) * 2] = hw;
        out[(blockIdx.x * 16 + wave) * 2 + 1] = xcc;
    }
    // stay resident long enough for the whole grid to be placed
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);
}
__global__ __launch_bounds__(512) void k_chain(float* out, unsigned* simd_of, int steps, unsigned mask, double f1, double f2, double f3) {
    __shared__ int first_on_simd[4], second_on_simd[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const int simd = (hw >> 4) & 3;
    if (threadIdx.x < 4) first_on_simd[threadIdx.x] = second_on_simd[threadIdx.x] = 99;
    __syncthreads();
    if (lane == 0) atomicMin(&first_on_simd[simd], wave);
    __syncthreads();
    if (lane == 0 && first_on_simd[simd] != wave) atomicMin(&second_on_simd[simd], wave);
    __syncthreads();
    // role r = 2 * simd + (0: first wavefront on that SIMD, 1: second); bit r of mask: run a chain
    const int rank = first_on_simd[simd] == wave ? 0 : second_on_simd[simd] == wave ? 1 : 2;
    if (lane == 0 && blockIdx.x == 0) simd_of[wave] = simd * 4 + rank;
    if (rank > 1 || !((mask >> (2 * simd + rank)) & 1u)) return;
    double v1 = 1 + lane, v2 = 1, v3 = 1;
    for (int s = 0; s < steps; ++s) {
        double v0 = (double)(float)(s & 7);
        v0 += v1 * f1;
        v0 += v2 * f2;
        v0 += v3 * f3;
        v3 = v2;
        v2 = v1;
        v1 = v0;
    }
    out[blockIdx.x * 512 + threadIdx.x] = (float)v1;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 4096 * 16 * 2 * 4);
    std::vector<unsigned> h(4096 * 16 * 2);
    for (int nw : {3, 5, 6, 7, 8}) {
        hipMemset(d, 0xff, 4096 * 16 * 2 * 4);
        const int nwg = 448;
        hipLaunchKernelGGL(k_where, dim3(nwg), dim3(nw * 64), 0, 0, d, 300);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        std::map<unsigned, std::vector<int>> cu_wgs;  // (xcc, se, sh, cu) -> workgroups
        for (int b = 0; b < nwg; ++b) {
            const unsigned hw = h[(b * 16) * 2], xcc = h[(b * 16) * 2 + 1] & 0xf;
            cu_wgs[(xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)].push_back(b);
        }
        int two = 0, clash0 = 0, clash01 = 0, shown = 0;
        for (auto& kv : cu_wgs) {
            if (kv.second.size() < 2) continue;
            ++two;
            int cnt0[4] = {0, 0, 0, 0}, cnt01[4] = {0, 0, 0, 0};
            for (int b : kv.second) {
                ++cnt0[(h[(b * 16 + 0) * 2] >> 4) & 3];
                ++cnt01[(h[(b * 16 + 0) * 2] >> 4) & 3];
                if (nw > 1) ++cnt01[(h[(b * 16 + 1) * 2] >> 4) & 3];
            }
            bool c0 = false, c01 = false;
            for (int s = 0; s < 4; ++s) c0 |= cnt0[s] > 1, c01 |= cnt01[s] > 1;
            clash0 += c0, clash01 += c01;
            if (shown < 3) {
                ++shown;
                printf("  %d wavefronts/workgroup, CU %06x:", nw, kv.first);
                for (int b : kv.second) {
                    printf(" wg %d SIMDs [", b);
                    for (int wv = 0; wv < nw; ++wv) printf("%u", (h[(b * 16 + wv) * 2] >> 4) & 3);
                    printf("]");
                }
                printf("\n");
            }
        }
        printf("%d wavefronts/workgroup, %d workgroups: %zu CUs used, %d with two or more workgroups; wave 0 of two workgroups on one SIMD on %d of them, "
               "waves 0/1 of two workgroups overlapping on %d\n", nw, nwg, cu_wgs.size(), two, clash0, clash01);
    }
    float* o;
    hipMalloc(&o, 256 * 512 * 4);
    unsigned* so;
    hipMalloc(&so, 64);
    const int steps = 200000;
    struct { const char* name; unsigned mask; } cases[] = {{"one chain per CU", 0x01}, {"two chains, SIMDs 0 and 1", 0x05}, {"two chains, both on SIMD 0", 0x03},
                                                          {"four chains, one per SIMD", 0x55}, {"eight chains, two per SIMD", 0xff}};
    for (auto& c : cases) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k_chain, dim3(256), dim3(512), 0, 0, o, so, 1000, c.mask, 1.2e-3, -3.1e-4, 7.7e-5);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_chain, dim3(256), dim3(512), 0, 0, o, so, steps, c.mask, 1.2e-3, -3.1e-4, 7.7e-5);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned hs[8];
        hipMemcpy(hs, so, 32, hipMemcpyDeviceToHost);
        printf("%-32s %.2f ns per step   (workgroup 0: wavefront -> SIMD*4+rank:", c.name, ms * 1e6 / steps);
        for (int i = 0; i < 8; ++i) printf(" %u", hs[i]);
        printf(")\n");
    }
    return 0;
}
