// VALU issue-rate microbenchmark on gfx950 (MI355X): how many cycles one SIMD needs per wave64
// vector instruction, at 1/2/4/8 waves per SIMD, for the instruction kinds the PreSync tile kernel
// (lmeds_kernel) is made of.  The loop bodies are inline asm (64 instructions per iteration, sixteen
// independent dependency chains), so what is timed is exactly what is written.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate tools/ubench/valu_rate.hip && ./valu_rate
//
// Output: one row per (instruction kind, waves per SIMD): wall time, the in-kernel shader clock
// (delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups) and
// cycles per wave-instruction per SIMD = clock x time / (instructions issued on one SIMD).
// MI355X_MICROARCH.md quotes "v_fma_f32 (wave64): 2 cyc (SIMD-32); one wave alone: 4"; the FP32
// vector peak of 157.3 TFLOP/s is 64 FLOP/clk/SIMD, which a wave64 v_fma_f32 (128 FLOP) reaches at
// 2 cycles and a wave64 v_pk_fma_f32 (256 FLOP) at 4.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP4(X) X X X X
#define FMA16                                                                                                       \
    asm volatile("v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n"             \
                 "v_fma_f32 %3, %3, %16, %17\n v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n"             \
                 "v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n v_fma_f32 %8, %8, %16, %17\n"             \
                 "v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"         \
                 "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n"       \
                 "v_fma_f32 %15, %15, %16, %17\n"                                                                     \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),     \
                   "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) \
                 : "v"(a), "v"(b));
#define MUL16                                                                                                       \
    asm volatile("v_mul_f32 %0, %0, %16\n v_mul_f32 %1, %1, %16\n v_mul_f32 %2, %2, %16\n"                            \
                 "v_mul_f32 %3, %3, %16\n v_mul_f32 %4, %4, %16\n v_mul_f32 %5, %5, %16\n"                            \
                 "v_mul_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n v_mul_f32 %8, %8, %16\n"                            \
                 "v_mul_f32 %9, %9, %16\n v_mul_f32 %10, %10, %16\n v_mul_f32 %11, %11, %16\n"                        \
                 "v_mul_f32 %12, %12, %16\n v_mul_f32 %13, %13, %16\n v_mul_f32 %14, %14, %16\n"                      \
                 "v_mul_f32 %15, %15, %16\n"                                                                          \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),     \
                   "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) \
                 : "v"(a));
#define MOV16                                                                                                       \
    asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"                        \
                 "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %8\n"                        \
                 "v_mov_b32 %8, %9\n v_mov_b32 %9, %10\n v_mov_b32 %10, %11\n v_mov_b32 %11, %12\n"                   \
                 "v_mov_b32 %12, %13\n v_mov_b32 %13, %14\n v_mov_b32 %14, %15\n v_mov_b32 %15, %0\n"                 \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),     \
                   "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
#define PKFMA16                                                                                                     \
    asm volatile("v_pk_fma_f32 %0, %0, %16, %17\n v_pk_fma_f32 %1, %1, %16, %17\n v_pk_fma_f32 %2, %2, %16, %17\n"    \
                 "v_pk_fma_f32 %3, %3, %16, %17\n v_pk_fma_f32 %4, %4, %16, %17\n v_pk_fma_f32 %5, %5, %16, %17\n"    \
                 "v_pk_fma_f32 %6, %6, %16, %17\n v_pk_fma_f32 %7, %7, %16, %17\n v_pk_fma_f32 %8, %8, %16, %17\n"    \
                 "v_pk_fma_f32 %9, %9, %16, %17\n v_pk_fma_f32 %10, %10, %16, %17\n v_pk_fma_f32 %11, %11, %16, %17\n" \
                 "v_pk_fma_f32 %12, %12, %16, %17\n v_pk_fma_f32 %13, %13, %16, %17\n v_pk_fma_f32 %14, %14, %16, %17\n" \
                 "v_pk_fma_f32 %15, %15, %16, %17\n"                                                                  \
                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]),     \
                   "+v"(p[8]), "+v"(p[9]), "+v"(p[10]), "+v"(p[11]), "+v"(p[12]), "+v"(p[13]), "+v"(p[14]), "+v"(p[15]) \
                 : "v"(va), "v"(vb));
#define FMA64_16                                                                                                    \
    asm volatile("v_fma_f64 %0, %0, %16, %17\n v_fma_f64 %1, %1, %16, %17\n v_fma_f64 %2, %2, %16, %17\n"             \
                 "v_fma_f64 %3, %3, %16, %17\n v_fma_f64 %4, %4, %16, %17\n v_fma_f64 %5, %5, %16, %17\n"             \
                 "v_fma_f64 %6, %6, %16, %17\n v_fma_f64 %7, %7, %16, %17\n v_fma_f64 %8, %8, %16, %17\n"             \
                 "v_fma_f64 %9, %9, %16, %17\n v_fma_f64 %10, %10, %16, %17\n v_fma_f64 %11, %11, %16, %17\n"         \
                 "v_fma_f64 %12, %12, %16, %17\n v_fma_f64 %13, %13, %16, %17\n v_fma_f64 %14, %14, %16, %17\n"       \
                 "v_fma_f64 %15, %15, %16, %17\n"                                                                     \
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),     \
                   "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15]) \
                 : "v"(da), "v"(db));

enum { M_FMA, M_MUL, M_MOV, M_PKFMA, M_FMA64, M_CMPCNT, M_STAGEC, M_STAGEC2, M_STAGEC4, M_COUNT };
const char* kNames[M_COUNT] = {"v_fma_f32", "v_mul_f32", "v_mov_b32", "v_pk_fma_f32", "v_fma_f64",
                               "v_cmp+s_bcnt1+s_add", "stageC(3 b128 + 6 pk_fma + 4 cmp/cnt)",
                               "stageC x2 hyp, count only (3 b128 + 12 pk_fma + 8 cmp/cnt)",
                               "stageC x4 hyp, count only (3 b128 + 24 pk_fma + 16 cmp/cnt)"};
// VALU instructions per loop iteration, per mode
const int kValuPerIter[M_COUNT] = {64, 64, 64, 64, 64, 64, 80, 160, 320};

struct Stamp { unsigned long long t0, t1, r0, r1; };

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, Stamp* stamps, int iters, float a, float b) {
    constexpr int kRows = (MODE == M_STAGEC || MODE == M_STAGEC2 || MODE == M_STAGEC4) ? 2048 : 4; // 24 KB only where it is used (occupancy)
    __shared__ __attribute__((aligned(16))) float tile[3][kRows];
    float x[16];
    v2f p[16];
    double d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float v = threadIdx.x * 1e-3f + i;
        x[i] = (MODE == M_PKFMA || MODE == M_FMA64) ? 0.f : v;
        p[i] = MODE == M_PKFMA ? v2f{v, v + 0.5f} : v2f{0.f, 0.f};
        d[i] = MODE == M_FMA64 ? (double)v : 0.0;
    }
    for (int i = threadIdx.x; i < 3 * kRows; i += 256) tile[0][i] = (float)(i % 977) * 1e-3f - 0.4f;
    __syncthreads();
    const v2f va = {a, a}, vb = {b, b};
    const double da = a, db = b;
    unsigned cnt = 0;
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == M_FMA) { REP4(FMA16) }
        if (MODE == M_MUL) { REP4(MUL16) }
        if (MODE == M_MOV) { REP4(MOV16) }
        if (MODE == M_PKFMA) { REP4(PKFMA16) }
        if (MODE == M_FMA64) { REP4(FMA64_16) }
        if (MODE == M_CMPCNT) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    cnt += (unsigned)__builtin_popcountll(__builtin_amdgcn_fcmpf(a, __builtin_fabsf(x[i]), 2));
            a += 1e-9f;
        }
        if (MODE == M_STAGEC) {
            // the inner loop of lmeds_kernel stage C for one hypothesis over a 2048-row tile:
            // per lane 8 x (3 ds_read_b128 + 6 v_pk_fma_f32 + 4 v_cmp/s_bcnt1/s_add) = 48 + 32 VALU
            const float4* p4x = (const float4*)tile[0];
            const float4* p4y = (const float4*)tile[1];
            const float4* p4z = (const float4*)tile[2];
            const int lane = threadIdx.x & 63;
            float r[32];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if ((m & 1) == 0) __builtin_amdgcn_sched_barrier(0);
                const int idx = m * 64 + lane;
                const float4 X = p4x[idx], Y = p4y[idx], Z = p4z[idx];
                const v2f r01 = v2f{X.x, X.y} * va.x + v2f{Y.x, Y.y} * b + v2f{Z.x, Z.y} * x[0];
                const v2f r23 = v2f{X.z, X.w} * va.x + v2f{Y.z, Y.w} * b + v2f{Z.z, Z.w} * x[0];
                r[4 * m] = r01.x; r[4 * m + 1] = r01.y; r[4 * m + 2] = r23.x; r[4 * m + 3] = r23.y;
            }
#pragma unroll
            for (int m = 0; m < 32; ++m)
                cnt += (unsigned)__builtin_popcountll(__builtin_amdgcn_fcmpf(a, __builtin_fabsf(r[m]), 2));
            a += 1e-9f;
            x[0] += (float)(cnt & 1u) * 1e-9f; // the next hypothesis depends on this one's count, as in the kernel
        }
        if (MODE == M_STAGEC2 || MODE == M_STAGEC4) {
            // one sweep of the tile for G hypotheses, counting only (no residual kept): the pre-filter form
            constexpr int G = MODE == M_STAGEC2 ? 2 : 4;
            const float4* p4x = (const float4*)tile[0];
            const float4* p4y = (const float4*)tile[1];
            const float4* p4z = (const float4*)tile[2];
            const int lane = threadIdx.x & 63;
            unsigned c[G];
#pragma unroll
            for (int q = 0; q < G; ++q) c[q] = 0;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if ((m & 1) == 0) __builtin_amdgcn_sched_barrier(0);
                const int idx = m * 64 + lane;
                const float4 X = p4x[idx], Y = p4y[idx], Z = p4z[idx];
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const v2f r01 = v2f{X.x, X.y} * x[3 * q] + v2f{Y.x, Y.y} * x[3 * q + 1] + v2f{Z.x, Z.y} * x[3 * q + 2];
                    const v2f r23 = v2f{X.z, X.w} * x[3 * q] + v2f{Y.z, Y.w} * x[3 * q + 1] + v2f{Z.z, Z.w} * x[3 * q + 2];
                    c[q] += (unsigned)__builtin_popcountll(__builtin_amdgcn_fcmpf(a, __builtin_fabsf(r01.x), 2));
                    c[q] += (unsigned)__builtin_popcountll(__builtin_amdgcn_fcmpf(a, __builtin_fabsf(r01.y), 2));
                    c[q] += (unsigned)__builtin_popcountll(__builtin_amdgcn_fcmpf(a, __builtin_fabsf(r23.x), 2));
                    c[q] += (unsigned)__builtin_popcountll(__builtin_amdgcn_fcmpf(a, __builtin_fabsf(r23.y), 2));
                }
            }
            a += 1e-9f;
#pragma unroll
            for (int q = 0; q < G; ++q) { cnt += c[q]; x[3 * q] += (float)(c[q] & 1u) * 1e-9f; }
        }
    }
    if (threadIdx.x == 0) {
        Stamp s;
        s.t0 = t0; s.r0 = r0;
        s.t1 = __builtin_amdgcn_s_memtime();
        s.r1 = __builtin_amdgcn_s_memrealtime();
        stamps[blockIdx.x] = s;
    }
    float acc = (float)cnt;
    // only the arrays the mode works on stay live (register budget: 8 waves per SIMD need <= 64 VGPRs)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (MODE == M_PKFMA) acc += p[i].x + p[i].y;
        else if (MODE == M_FMA64) acc += (float)d[i];
        else acc += x[i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
void run(int waves_per_simd, FILE* csv) {
    const int grid = 256 * waves_per_simd; // 256 CUs, one 4-wave workgroup per CU per wave-per-SIMD
    int max_blocks = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&max_blocks, k<MODE>, 256, 0);
    if (waves_per_simd > max_blocks) return; // the workgroups of a CU would not all be resident
    float* out;
    Stamp* st;
    hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    hipMalloc(&st, (size_t)grid * sizeof(Stamp));
    const int iters = MODE == M_STAGEC4 ? 1000 : MODE == M_STAGEC2 ? 2000 : (MODE == M_STAGEC || MODE == M_CMPCNT) ? 4000 : 8000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, st, 200, 1.0001f, 0.5f); // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, st, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(grid);
    hipMemcpy(h.data(), st, (size_t)grid * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (auto& s : h) {
        const double dt = (double)(s.t1 - s.t0), dr = (double)(s.r1 - s.r0);
        if (dr > 0) ghz.push_back(dt / dr * 0.1); // s_memrealtime ticks at 100 MHz
        cyc.push_back(dt);
    }
    std::sort(ghz.begin(), ghz.end());
    std::sort(cyc.begin(), cyc.end());
    const double clk = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
    const double wg_cycles = cyc[cyc.size() / 2];
    const double valu_per_simd = (double)iters * kValuPerIter[MODE] * waves_per_simd;
    // two views: (1) in-kernel cycles of the median workgroup / instructions issued on its SIMD while it ran
    // (all waves_per_simd workgroups of a CU run concurrently), (2) wall time x clock / instructions per SIMD
    const double cyc_in_kernel = wg_cycles / valu_per_simd;
    const double cyc_wall = clk * 1e9 * ms * 1e-3 / valu_per_simd;
    printf("%-40s waves/SIMD %d  %8.3f ms  clock %.3f GHz  cycles/VALU-instr/SIMD: %.2f (in-kernel) %.2f (wall)\n",
           kNames[MODE], waves_per_simd, ms, clk, cyc_in_kernel, cyc_wall);
    if (csv) fprintf(csv, "\"%s\",%d,%.4f,%.4f,%.3f,%.3f\n", kNames[MODE], waves_per_simd, ms, clk, cyc_in_kernel, cyc_wall);
    hipFree(out);
    hipFree(st);
}

int main(int argc, char** argv) {
    FILE* csv = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (csv) fprintf(csv, "kind,waves_per_simd,ms,clock_ghz,cycles_per_valu_in_kernel,cycles_per_valu_wall\n");
    for (int w : {1, 2, 3, 4, 5, 6, 8}) {
        run<M_FMA>(w, csv);
        run<M_MUL>(w, csv);
        run<M_MOV>(w, csv);
        run<M_PKFMA>(w, csv);
        run<M_FMA64>(w, csv);
        run<M_CMPCNT>(w, csv);
        run<M_STAGEC>(w, csv);
        run<M_STAGEC2>(w, csv);
        run<M_STAGEC4>(w, csv);
    }
    if (csv) fclose(csv);
    return 0;
}
