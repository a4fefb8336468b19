// stagec_mfma.hip -- one bounded experiment: would the matrix pipe pay for stage C of the PreSync
// tile kernel (kernels/lmeds.hpp)?  Stage C's first pass over a hypothesis is "residuals r = n . v for all N rows, count
// |r| < T" -- an N x 3 . 3 x 20 contraction followed by a count per column.  The product does it on the VALU
// (sweep_tile: v_pk_fma_f32 on ds_read_b128 operands) and counts with v_cmp + s_bcnt1 + s_add.  The alternative:
// v_mfma_f32_16x16x4_f32 (K padded 3 -> 4, the 20 hypotheses padded to 2 x 16 columns; exact f32, bitwise the fmaf
// chain sweep_tile issues), counting per lane with v_cmp + v_addc against the same bound.
//
// This file times exactly those two inner loops in the kernel's own geometry -- 256-thread workgroups, five per CU, a
// 2048-row unit-vector tile in LDS (SoA), 20 hypothesis directions, one bound per candidate -- and checks that both
// produce the same counts.  What it leaves out favours the MFMA form: the product's sequential tightening of the
// bound (a later hypothesis is rejected against a smaller T; here every hypothesis meets the same T), its early
// rejection before the last group of rows, and the re-sweep the MFMA form would need for every hypothesis that
// passes (its residuals are not in the registers of one wave).
//
//   hipcc --offload-arch=gfx950 -O3 -o stagec_mfma tools/ubench/stagec_mfma.hip && ./stagec_mfma
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(2); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kRows = 2048, kHyp = 20, kBlock = 256, kCand = 32;

struct Params {
    const float* tile;   // [3][kRows] unit rows
    const float* hyp;    // [kCand][kHyp][4]
    const float* bound;  // [kCand]
    uint32_t* counts;    // [grid][kCand][32]
};

// (kernels/lmeds.hpp: lane 0 alone performs the atomic, written as one asm statement -- hipcc's structuriser turns the
// obvious `if (lane == 0) j = atomicAdd(..); j = readfirstlane(j);` inside a loop into a per-lane waterfall that never
// terminates; this file's first version hung on exactly that until its timeout)
__device__ __forceinline__ uint32_t wave_pop(uint32_t* counter) {
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)counter;
    const uint32_t one = 1u;
    uint32_t old;
    unsigned long long save;
    asm volatile("s_mov_b64 %1, exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "ds_add_rtn_u32 %0, %2, %3\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b64 exec, %1"
                 : "=&v"(old), "=&s"(save)
                 : "v"(addr), "v"(one)
                 : "memory");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
}

// ---- the product's form: a wave takes a hypothesis, holds all 2048 residuals in 32 registers, counts on the scalar unit
__global__ __launch_bounds__(kBlock, 5) void valu_kernel(Params p) {
    __shared__ __attribute__((aligned(16))) float s_n[3][kRows];
    __shared__ f4 s_hyp[kHyp];
    __shared__ uint32_t s_next;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 3 * kRows; i += kBlock) (&s_n[0][0])[i] = p.tile[i];
    const f4* p4x = reinterpret_cast<const f4*>(s_n[0]);
    const f4* p4y = reinterpret_cast<const f4*>(s_n[1]);
    const f4* p4z = reinterpret_cast<const f4*>(s_n[2]);
    for (int c = 0; c < kCand; ++c) {
        __syncthreads();
        if (tid < kHyp) s_hyp[tid] = reinterpret_cast<const f4*>(p.hyp)[c * kHyp + tid];
        if (tid == 0) s_next = 0;
        __syncthreads();
        const float T = p.bound[c];
        for (int guard = 0; guard <= kHyp; ++guard) { // (bounded: a wave pops at most kHyp + 1 times)
            const uint32_t h = wave_pop(&s_next);
            if (h >= (uint32_t)kHyp) break;
            const f4 hv = s_hyp[h];
            uint32_t cnt = 0;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if ((m & 1) == 0) __builtin_amdgcn_sched_barrier(0);
                const int idx = m * 64 + lane;
                const f4 x = p4x[idx], y = p4y[idx], z = p4z[idx];
                const v2f r01 = v2f{x.x, x.y} * hv.x + v2f{y.x, y.y} * hv.y + v2f{z.x, z.y} * hv.z;
                const v2f r23 = v2f{x.z, x.w} * hv.x + v2f{y.z, y.w} * hv.y + v2f{z.z, z.w} * hv.z;
                cnt += (uint32_t)__builtin_popcountll(__builtin_amdgcn_fcmpf(T, fabsf(r01.x), 2));
                cnt += (uint32_t)__builtin_popcountll(__builtin_amdgcn_fcmpf(T, fabsf(r01.y), 2));
                cnt += (uint32_t)__builtin_popcountll(__builtin_amdgcn_fcmpf(T, fabsf(r23.x), 2));
                cnt += (uint32_t)__builtin_popcountll(__builtin_amdgcn_fcmpf(T, fabsf(r23.y), 2));
            }
            if (lane == 0) p.counts[((size_t)blockIdx.x * kCand + c) * 32 + h] = cnt;
        }
    }
}

// ---- the matrix-pipe form: a wave takes 32 blocks of 16 rows, two MFMAs per block (hypotheses 0-15, 16-31), counts per lane
__global__ __launch_bounds__(kBlock, 5) void mfma_kernel(Params p) {
    __shared__ __attribute__((aligned(16))) float s_n[3][kRows]; // (the padded K = 3 is a zero operand, not a plane)
    __shared__ uint32_t s_cnt[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * kRows; i += kBlock) (&s_n[0][0])[i] = p.tile[i];
    const int k = lane >> 4, col = lane & 15;
    for (int c = 0; c < kCand; ++c) {
        __syncthreads();
        if (tid < 32) s_cnt[tid] = 0;
        // B[k][col]: component k of hypothesis col (column block 0) / 16 + col (block 1); zero beyond kHyp and for k = 3
        const float* hc = p.hyp + (size_t)c * kHyp * 4;
        const float b0 = (k < 3) ? hc[col * 4 + k] : 0.f;
        const float b1 = (k < 3 && 16 + col < kHyp) ? hc[(16 + col) * 4 + k] : 0.f;
        const float T = p.bound[c];
        __syncthreads();
        uint32_t cnt0 = 0, cnt1 = 0;
#pragma unroll 2
        for (int rb = 0; rb < 32; ++rb) {
            const int row = (wave * 32 + rb) * 16 + col;
            const float a = k < 3 ? s_n[k][row] : 0.f;
            const f4 zero = {0.f, 0.f, 0.f, 0.f};
            const f4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, zero, 0, 0, 0);
            const f4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, zero, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                cnt0 += (T > fabsf(d0[i])) ? 1u : 0u;
                cnt1 += (T > fabsf(d1[i])) ? 1u : 0u;
            }
        }
        // the four lane groups hold different rows of the same column: add them, then the waves
        cnt0 += __shfl_xor((int)cnt0, 16); cnt0 += __shfl_xor((int)cnt0, 32);
        cnt1 += __shfl_xor((int)cnt1, 16); cnt1 += __shfl_xor((int)cnt1, 32);
        if (lane < 16) { atomicAdd(&s_cnt[lane], cnt0); atomicAdd(&s_cnt[16 + lane], cnt1); }
        __syncthreads();
        if (tid < kHyp) p.counts[((size_t)blockIdx.x * kCand + c) * 32 + tid] = s_cnt[tid];
    }
}

int main() {
    int n_cu = 256;
    CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    const int grid = n_cu * 5 * 4;
    std::vector<float> tile(3 * kRows), hyp((size_t)kCand * kHyp * 4), bound(kCand);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
    for (int r = 0; r < kRows; ++r) {
        float x = rnd(), y = rnd(), z = rnd(), n = std::sqrt(x * x + y * y + z * z) + 1e-9f;
        tile[r] = x / n; tile[kRows + r] = y / n; tile[2 * kRows + r] = z / n;
    }
    for (size_t i = 0; i < (size_t)kCand * kHyp; ++i) {
        float x = rnd(), y = rnd(), z = rnd(), n = std::sqrt(x * x + y * y + z * z) + 1e-9f;
        hyp[4 * i] = x / n; hyp[4 * i + 1] = y / n; hyp[4 * i + 2] = z / n; hyp[4 * i + 3] = 0.f;
    }
    for (int c = 0; c < kCand; ++c) bound[c] = 0.2f + 0.01f * c; // |n . v| < T for ~20-50 % of uniformly random rows
    float *d_tile, *d_hyp, *d_bound;
    uint32_t *d_ca, *d_cb;
    const size_t cbytes = (size_t)grid * kCand * 32 * 4;
    CK(hipMalloc(&d_tile, tile.size() * 4)); CK(hipMalloc(&d_hyp, hyp.size() * 4)); CK(hipMalloc(&d_bound, bound.size() * 4));
    CK(hipMalloc(&d_ca, cbytes)); CK(hipMalloc(&d_cb, cbytes));
    CK(hipMemcpy(d_tile, tile.data(), tile.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_hyp, hyp.data(), hyp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bound, bound.data(), bound.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_ca, 0, cbytes)); CK(hipMemset(d_cb, 0, cbytes));
    Params pa{d_tile, d_hyp, d_bound, d_ca}, pb{d_tile, d_hyp, d_bound, d_cb};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_a = 1e30f, ms_b = 1e30f;
    for (int rep = 0; rep < 5; ++rep) { // interleaved, best of five
        float ms;
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(valu_kernel, dim3(grid), dim3(kBlock), 0, 0, pa);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        ms_a = ms < ms_a ? ms : ms_a;
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(mfma_kernel, dim3(grid), dim3(kBlock), 0, 0, pb);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        ms_b = ms < ms_b ? ms : ms_b;
    }
    CK(hipGetLastError());
    std::vector<uint32_t> ca(cbytes / 4), cb(cbytes / 4);
    CK(hipMemcpy(ca.data(), d_ca, cbytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cb.data(), d_cb, cbytes, hipMemcpyDeviceToHost));
    size_t diff = 0, total = 0;
    for (int b = 0; b < grid; ++b)
        for (int c = 0; c < kCand; ++c)
            for (int h = 0; h < kHyp; ++h) { ++total; diff += ca[((size_t)b * kCand + c) * 32 + h] != cb[((size_t)b * kCand + c) * 32 + h]; }
    const double pairs = (double)grid * kCand; // (workgroup, candidate) pairs = what one (frame, candidate) costs in stage C's first pass
    printf("stage C first pass, %d workgroups x %d candidates, %d rows x %d hypotheses, five workgroups per CU\n", grid, kCand, kRows, kHyp);
    printf("  VALU (v_pk_fma_f32 + v_cmp + s_bcnt1, the product's form)   %8.3f ms   %7.1f ns per (tile, candidate)\n", ms_a, ms_a * 1e6 / pairs * n_cu * 5);
    printf("  MFMA (v_mfma_f32_16x16x4_f32 + per-lane v_cmp / v_addc)     %8.3f ms   %7.1f ns per (tile, candidate)\n", ms_b, ms_b * 1e6 / pairs * n_cu * 5);
    printf("  ratio MFMA / VALU %.3f;  counts that differ: %zu of %zu (sample: %u vs %u)\n", ms_b / ms_a, diff, total, ca[0], cb[0]);
    return diff ? 1 : 0;
}
