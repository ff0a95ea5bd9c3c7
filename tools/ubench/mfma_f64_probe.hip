// fp64 MFMA probe (development tool): (1) is v_mfma_f64_4x4x4 a chain of correctly rounded FMAs in k order, and which lane holds
// which element; (2) what it costs per SIMD, alone and beside fp64 VALU work of other waves.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/ubench/mfma_f64_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void one_mfma(const double* a, const double* b, const double* c, double* d, double* d16) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc = {c[l], c[l + 64], c[l + 128], c[l + 192]};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], acc, 0, 0, 0);
    d16[l] = acc[0]; d16[l + 64] = acc[1]; d16[l + 128] = acc[2]; d16[l + 192] = acc[3];
}

template <int KIND>   // 0: MFMA 4x4x4 only, 1: VALU fma only, 2: waves alternate by workgroup parity, 3: MFMA 16x16x4 only, 4: each wave both
__global__ __launch_bounds__(64) void rate(double* out, int iters, double seed) {
    double a = seed + threadIdx.x, b = seed * 0.5;
    double m0 = 0.0, m1 = 1.0, m2 = 2.0, m3 = 3.0;
    double v0 = 0.5, v1 = 1.5, v2 = 2.5, v3 = 3.5, v4 = 4.5, v5 = 5.5, v6 = 6.5, v7 = 7.5;
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 w0 = {0, 0, 0, 0}, w1 = {1, 1, 1, 1};
    const bool do_m = KIND == 0 || KIND == 4 || (KIND == 2 && (blockIdx.x & 1) == 0);
    const bool do_v = KIND == 1 || KIND == 4 || (KIND == 2 && (blockIdx.x & 1) == 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (KIND == 3) {
                w0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w0, 0, 0, 0);
                w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w1, 0, 0, 0);
            }
            if (do_m) {
                m0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m0, 0, 0, 0);
                m1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m1, 0, 0, 0);
                m2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m2, 0, 0, 0);
                m3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m3, 0, 0, 0);
            }
            if (do_v) {
                v0 = __builtin_fma(a, b, v0); v1 = __builtin_fma(a, b, v1); v2 = __builtin_fma(a, b, v2); v3 = __builtin_fma(a, b, v3);
                v4 = __builtin_fma(a, b, v4); v5 = __builtin_fma(a, b, v5); v6 = __builtin_fma(a, b, v6); v7 = __builtin_fma(a, b, v7);
                v0 = __builtin_fma(a, b, v0); v1 = __builtin_fma(a, b, v1); v2 = __builtin_fma(a, b, v2); v3 = __builtin_fma(a, b, v3);
                v4 = __builtin_fma(a, b, v4); v5 = __builtin_fma(a, b, v5); v6 = __builtin_fma(a, b, v6); v7 = __builtin_fma(a, b, v7);
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = m0 + m1 + m2 + m3 + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + w0[0] + w1[1];
}

template <int KIND>
static void run_rate(const char* name, int w, int ncu) {
    const int grid = ncu * 4 * w, iters = 256;
    double* out; (void)hipMalloc(&out, (size_t)grid * 64 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rate<KIND><<<grid, 64>>>(out, 2, 1e-9); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); rate<KIND><<<grid, 64>>>(out, iters, 1e-9); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: w waves x iters x 16 rounds
    printf("%-40s waves/SIMD %d: %.1f cycles@2.4GHz per round per wave-slot (round = 4 mfma4x4x4 | 16 v_fma_f64 | 2 mfma16x16x4)\n", name, w,
           ms * 1e-3 * 2.4e9 / ((double)iters * 16 * w));
    (void)hipFree(out);
}

int main() {
    std::vector<double> a(64), b(64), c(256), d(64), d16(256);
    srand(7);
    auto rnd = []() { return (rand() / (double)RAND_MAX - 0.5) * std::exp((rand() % 40) - 20.0) + rand() * 1e-25; };
    for (auto& x : a) x = rnd();
    for (auto& x : b) x = rnd();
    for (auto& x : c) x = rnd();
    double *da, *db, *dc, *dd, *dd16;
    (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dc, 2048); (void)hipMalloc(&dd, 512); (void)hipMalloc(&dd16, 2048);
    (void)hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
    (void)hipMemcpy(dc, c.data(), 2048, hipMemcpyHostToDevice);
    one_mfma<<<1, 64>>>(da, db, dc, dd, dd16);
    (void)hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
    (void)hipMemcpy(d16.data(), dd16, 2048, hipMemcpyDeviceToHost);
    // 4x4x4, 4 blocks: search the lane maps and the accumulation order
    int found = 0;
    for (int am = 0; am < 2; ++am) for (int bm = 0; bm < 2; ++bm) for (int dm = 0; dm < 2; ++dm) for (int ord = 0; ord < 4; ++ord) {
        int ok = 0;
        for (int blk = 0; blk < 4; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            const int ld = 16 * blk + (dm ? 4 * j + i : 4 * i + j);
            double acc = (ord & 2) ? 0.0 : c[ld];
            for (int kk = 0; kk < 4; ++kk) {
                const int k = (ord & 1) ? 3 - kk : kk;
                const double av = a[16 * blk + (am ? 4 * i + k : 4 * k + i)], bv = b[16 * blk + (bm ? 4 * j + k : 4 * k + j)];
                acc = std::fma(av, bv, acc);
            }
            if (ord & 2) acc = acc + c[ld];
            ok += acc == d[ld];
        }
        if (ok >= 60) { printf("4x4x4: A lane=16b+%s, B lane=16b+%s, D lane=16b+%s, k %s, C %s: %d/64 bit-equal\n", am ? "4i+k" : "4k+i", bm ? "4j+k" : "4k+j",
                               dm ? "4j+i" : "4i+j", (ord & 1) ? "descending" : "ascending", (ord & 2) ? "added last" : "first", ok); ++found; }
    }
    if (!found) printf("4x4x4: no map / order gives bit-equal results (not a chain of FMAs in k order?)\n");
    // 16x16x4: A[l&15][k=l>>4], B[k=l>>4][l&15], D col=l&15 row=(l>>4)+4*reg
    for (int ord = 0; ord < 4; ++ord) {
        int ok = 0;
        for (int reg = 0; reg < 4; ++reg) for (int l = 0; l < 64; ++l) {
            const int row = (l >> 4) + 4 * reg, col = l & 15;
            double acc = (ord & 2) ? 0.0 : c[l + 64 * reg];
            for (int kk = 0; kk < 4; ++kk) {
                const int k = (ord & 1) ? 3 - kk : kk;
                acc = std::fma(a[row + 16 * k], b[col + 16 * k], acc);
            }
            if (ord & 2) acc = acc + c[l + 64 * reg];
            ok += acc == d16[l + 64 * reg];
        }
        printf("16x16x4: k %s, C %s: %d/256 bit-equal\n", (ord & 1) ? "descending" : "ascending", (ord & 2) ? "added last" : "first", ok);
    }
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    for (int w : {1, 2, 4, 8}) {
        run_rate<0>("mfma_f64_4x4x4 only", w, ncu);
        run_rate<3>("mfma_f64_16x16x4 only", w, ncu);
        run_rate<1>("v_fma_f64 only", w, ncu);
        run_rate<2>("half the waves mfma4, half v_fma", w, ncu);
        run_rate<4>("every wave mfma4 + v_fma", w, ncu);
    }
    return 0;
}
