// Issue-rate probe for gfx950 (development tool): cycles per wave64 VALU instruction on one SIMD, by instruction kind and
// by waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/ubench/issue_rate.hip -o /tmp/issue_rate && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 512
template <int KIND>
__global__ __launch_bounds__(64) void probe(double* out, long long* cyc, double s0, double s1, int iters) {
    double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    const double v0 = s0 + threadIdx.x, v1 = s1 - threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (KIND == 0) {          // v_fma_f64, vector operands
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(v0), "v"(v1));
            } else if (KIND == 1) {   // v_fma_f64 with one scalar operand
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s0), "v"(v1));
            } else if (KIND == 2) {   // v_mul_f64
                asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                             "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(v0));
            } else if (KIND == 3) {   // v_add_u32
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(i0));
            } else if (KIND == 4) {   // v_cndmask_b32 (vcc)
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(i0) : "vcc");
            } else if (KIND == 5) {   // v_fma_f64 dependent chain (latency)
                asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                             "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2"
                             : "+v"(a0) : "v"(v0), "v"(v1));
            } else if (KIND == 6) {   // v_mov_b32 dpp
                asm volatile("v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(i0));
            } else if (KIND == 7) {   // s_add_u32 (scalar)
                int x = it;
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                             "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(x) : : "scc");
                i0 += x & 0;
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int waves_per_simd, int ncu) {
    const int grid = ncu * 4 * waves_per_simd, iters = 64;
    double* out; long long* cyc;
    hipMalloc(&out, (size_t)grid * 64 * 8); hipMalloc(&cyc, (size_t)grid * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<grid, 64>>>(out, cyc, 1.0000001, 1e-9, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND><<<grid, 64>>>(out, cyc, 1.0000001, 1e-9, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, (size_t)grid * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= grid;
    const double n = (double)iters * REP;
    // s_memtime ticks at the shader clock (guide); per-SIMD cost of one wave-instruction = wave cycles / instructions / waves sharing the SIMD
    printf("%-22s waves/SIMD %d: %.2f memtime ticks per instr per wave -> %.2f per SIMD slot; kernel %.3f ms -> %.2f cycles@2.4GHz per instr per SIMD\n",
           name, waves_per_simd, avg / n, avg / n / waves_per_simd, ms, ms * 1e-3 * 2.4e9 / (n * waves_per_simd));
    hipFree(out); hipFree(cyc);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, ncu, p.clockRate);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f64 vvv", w, ncu);
        run<1>("v_fma_f64 v,s,v", w, ncu);
        run<2>("v_mul_f64", w, ncu);
        run<3>("v_add_u32", w, ncu);
        run<4>("v_cndmask_b32", w, ncu);
        run<6>("v_mov_b32 dpp", w, ncu);
        run<7>("s_add_u32", w, ncu);
    }
    run<5>("v_fma_f64 dependent", 1, ncu);
    return 0;
}
