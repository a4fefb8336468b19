// VALU issue-rate microbenchmark on gfx950: v_fma_f32 vs v_pk_fma_f32 vs v_cmp+s_bcnt1, at 1..8 waves/SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x2}, p5 = {x3, x4}, p6 = {x5, x6}, p7 = {x7, x0};
    v2f va = {a, a}, vb = {b, b};
    unsigned cnt = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                p0 = p0 * va + vb; p1 = p1 * va + vb; p2 = p2 * va + vb; p3 = p3 * va + vb;
                p4 = p4 * va + vb; p5 = p5 * va + vb; p6 = p6 * va + vb; p7 = p7 * va + vb;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                cnt += __builtin_popcountll(__builtin_amdgcn_fcmpf(a, x0, 2)); cnt += __builtin_popcountll(__builtin_amdgcn_fcmpf(a, x1, 2));
                cnt += __builtin_popcountll(__builtin_amdgcn_fcmpf(a, x2, 2)); cnt += __builtin_popcountll(__builtin_amdgcn_fcmpf(a, x3, 2));
                cnt += __builtin_popcountll(__builtin_amdgcn_fcmpf(a, x4, 2)); cnt += __builtin_popcountll(__builtin_amdgcn_fcmpf(a, x5, 2));
                cnt += __builtin_popcountll(__builtin_amdgcn_fcmpf(a, x6, 2)); cnt += __builtin_popcountll(__builtin_amdgcn_fcmpf(a, x7, 2));
                a += 1e-9f;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + cnt;
}
template <int MODE>
void run(const char* name, int blocks_per_cu) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
    int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst = (double)iters * 64 * grid * 4; // wave-instructions
    printf("%-14s %d blocks/CU (%d waves/SIMD): %.3f ms  %.1f wave-instr/us/SIMD -> %.2f cycles per instr per SIMD at 2.4GHz\n", name, blocks_per_cu,
           blocks_per_cu, ms, inst / (ms * 1e3) / 1024, 2400.0 / (inst / (ms * 1e3) / 1024));
    hipFree(out);
}
int main() {
    for (int b : {1, 2, 4, 8}) { run<0>("v_fma_f32", b); run<1>("v_pk_fma_f32", b); run<2>("v_cmp+s_bcnt", b); }
    return 0;
}
