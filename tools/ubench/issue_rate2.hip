// Issue-rate probe II (development tool): more instruction kinds, 8 waves per SIMD, wall-clock cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 512
#define G8(ins) ins(0) ins(1) ins(2) ins(3) ins(4) ins(5) ins(6) ins(7)
template <int KIND>
__global__ __launch_bounds__(64) void probe(int* out, int iters, int sv) {
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    int j = threadIdx.x * 3 + sv;
    unsigned long long m = sv ? 0xF0F0F0F0F0F0F0F0ull : 1ull;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (KIND == 0) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(j) : "vcc");
            if (KIND == 1) asm volatile("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(j), "s"(m));
            if (KIND == 2) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(j) : "vcc");
            if (KIND == 3) asm volatile("v_and_or_b32 %0, %0, %8, %8\n v_and_or_b32 %1, %1, %8, %8\n v_and_or_b32 %2, %2, %8, %8\n v_and_or_b32 %3, %3, %8, %8\n v_and_or_b32 %4, %4, %8, %8\n v_and_or_b32 %5, %5, %8, %8\n v_and_or_b32 %6, %6, %8, %8\n v_and_or_b32 %7, %7, %8, %8" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(j));
            if (KIND == 4) asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(j));
            if (KIND == 5) asm volatile("v_bfe_u32 %0, %0, 20, 11\n v_bfe_u32 %1, %1, 20, 11\n v_bfe_u32 %2, %2, 20, 11\n v_bfe_u32 %3, %3, 20, 11\n v_bfe_u32 %4, %4, 20, 11\n v_bfe_u32 %5, %5, 20, 11\n v_bfe_u32 %6, %6, 20, 11\n v_bfe_u32 %7, %7, 20, 11" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(j));
            if (KIND == 6) asm volatile("v_lshlrev_b32 %0, 5, %0\n v_lshlrev_b32 %1, 5, %1\n v_lshlrev_b32 %2, 5, %2\n v_lshlrev_b32 %3, 5, %3\n v_lshlrev_b32 %4, 5, %4\n v_lshlrev_b32 %5, 5, %5\n v_lshlrev_b32 %6, 5, %6\n v_lshlrev_b32 %7, 5, %7" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(j));
            if (KIND == 7) asm volatile("v_min_i32 %0, %0, %8\n v_min_i32 %1, %1, %8\n v_min_i32 %2, %2, %8\n v_min_i32 %3, %3, %8\n v_min_i32 %4, %4, %8\n v_min_i32 %5, %5, %8\n v_min_i32 %6, %6, %8\n v_min_i32 %7, %7, %8" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(j));
            if (KIND == 8) asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %0, %1\n v_mov_b64 %0, %1\n v_mov_b64 %0, %1\n v_mov_b64 %0, %1\n v_mov_b64 %0, %1\n v_mov_b64 %0, %1\n v_mov_b64 %0, %1" : "+v"(m) : "v"(m));
            if (KIND == 9) asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %0, %0, %0\n v_add_f64 %0, %0, %0\n v_add_f64 %0, %0, %0\n v_add_f64 %0, %0, %0\n v_add_f64 %0, %0, %0\n v_add_f64 %0, %0, %0\n v_add_f64 %0, %0, %0" : "+v"(m));
            if (KIND == 10) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_add_u32 %1, %1, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_u32 %3, %3, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_add_u32 %5, %5, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_add_u32 %7, %7, %8" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(j) : "vcc");
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7 + (int)m;
}
template <int KIND>
void run(const char* name, int w, int ncu) {
    const int grid = ncu * 4 * w, iters = 64;
    int* out; (void)hipMalloc(&out, (size_t)grid * 64 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<KIND><<<grid, 64>>>(out, 2, 1); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); probe<KIND><<<grid, 64>>>(out, iters, 1); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s waves/SIMD %d: %.2f cycles@2.4GHz per instr per SIMD\n", name, w, ms * 1e-3 * 2.4e9 / ((double)iters * REP * w));
    (void)hipFree(out);
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    for (int w : {2, 8}) {
        run<0>("v_cndmask_b32 vcc", w, ncu); run<1>("v_cndmask_b32 sgpr mask", w, ncu); run<2>("v_cmp_lt_u32", w, ncu);
        run<3>("v_and_or_b32", w, ncu); run<4>("v_mov_b32", w, ncu); run<5>("v_bfe_u32", w, ncu); run<6>("v_lshlrev_b32", w, ncu);
        run<7>("v_min_i32", w, ncu); run<8>("v_mov_b64 (dependent)", w, ncu); run<9>("v_add_f64 (dependent)", w, ncu);
        run<10>("cndmask/add alternating", w, ncu);
    }
    return 0;
}
