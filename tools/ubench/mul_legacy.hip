// v_mul_legacy_f32 on gfx950: 0 x anything = 0 (the DX9 rule)?  Everything else as v_mul_f32, BIT FOR BIT -- denormal
// operands and results, infinities, NaNs included?   hipcc --offload-arch=gfx950 -O2 mul_legacy.hip -o mul_legacy && ./mul_legacy
// (kernels/common.hpp: mul_zero_wins, stage D of the LMedS kernels: a row's norm -- 0 for the rows beyond the frame -- times a
// dot product that is NaN for exactly those rows.)  Both products are computed ON THE DEVICE, by the two instructions, in the
// kernels' own floating-point mode; the comparison is of bit patterns.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const float* a, const float* b, float* legacy, float* plain, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r, q;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a[i]), "v"(b[i]));
    asm("v_mul_f32 %0, %1, %2" : "=v"(q) : "v"(a[i]), "v"(b[i]));
    legacy[i] = r;
    plain[i] = q;
}
static float from_bits(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
static uint32_t bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
int main() {
    const float nan = std::nanf(""), inf = INFINITY, dmin = from_bits(1u), dmax = from_bits(0x007fffffu);
    struct Cat { const char* name; size_t begin, end; };
    std::vector<float> a, b;
    std::vector<Cat> cats;
    auto open_cat = [&](const char* name) { cats.push_back({name, a.size(), a.size()}); };
    auto close_cat = [&]() { cats.back().end = a.size(); };
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    auto rnd_unit = [&]() { return from_bits((rnd() >> 9) | 0x3f800000u) - 1.5f; }; // [-0.5, 0.5)
    open_cat("a zero operand (the rule itself): 0 x {NaN, inf, -inf, 5, denormal}, -0 x NaN, NaN x 0, inf x 0");
    for (float z : {0.f, -0.f})
        for (float y : {nan, inf, -inf, 5.f, dmin, -dmax}) { a.push_back(z); b.push_back(y); a.push_back(y); b.push_back(z); }
    close_cat();
    open_cat("ordinary pairs, O(1e3) x O(1e-3)");
    for (int i = 0; i < 200000; ++i) { a.push_back(rnd_unit() * 1e3f); b.push_back(rnd_unit() * 1e-3f); }
    close_cat();
    open_cat("any two bit patterns (NaNs, infinities, denormals, both signs)");
    for (int i = 0; i < 400000; ++i) { a.push_back(from_bits(rnd())); b.push_back(from_bits(rnd())); }
    close_cat();
    open_cat("a denormal operand x a normal one");
    for (int i = 0; i < 100000; ++i) { a.push_back(from_bits((rnd() & 0x807fffffu))); b.push_back(rnd_unit() * 8.f); }
    close_cat();
    open_cat("normal operands, denormal (or underflowing) result");
    for (int i = 0; i < 100000; ++i) { a.push_back(rnd_unit() * 1e-20f); b.push_back(rnd_unit() * 1e-19f * (1.f + (rnd() & 1023))); }
    close_cat();
    open_cat("inf x finite nonzero, inf x inf, NaN x finite");
    for (int i = 0; i < 50000; ++i) {
        const float f = rnd_unit() * 7.f + (rnd_unit() >= 0 ? 10.f : -10.f);
        a.push_back((i & 1) ? inf : -inf); b.push_back(f);
        a.push_back(f); b.push_back((i & 2) ? inf : -inf);
        a.push_back(inf); b.push_back((i & 1) ? inf : -inf);
        a.push_back(nan); b.push_back(f);
    }
    close_cat();
    open_cat("overflowing products");
    for (int i = 0; i < 50000; ++i) { a.push_back(rnd_unit() * 1e30f); b.push_back(rnd_unit() * 1e30f); }
    close_cat();
    const int n = (int)a.size();
    float *da, *db, *dl, *dp;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dl, n * 4); hipMalloc(&dp, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dl, dp, n);
    std::vector<float> l(n), q(n);
    hipMemcpy(l.data(), dl, n * 4, hipMemcpyDeviceToHost); hipMemcpy(q.data(), dp, n * 4, hipMemcpyDeviceToHost);
    int total_bad = 0;
    for (const Cat& c : cats) {
        size_t zero_ops = 0, zero_rule_ok = 0, other = 0, other_same = 0, other_same_mod_nan = 0;
        for (size_t i = c.begin; i < c.end; ++i) {
            const bool zero = a[i] == 0.f || b[i] == 0.f;
            if (zero) { ++zero_ops; zero_rule_ok += (l[i] == 0.f) ? 1 : 0; continue; }
            ++other;
            const bool same = bits(l[i]) == bits(q[i]);
            other_same += same ? 1 : 0;
            other_same_mod_nan += (same || (std::isnan(l[i]) && std::isnan(q[i]))) ? 1 : 0; // (two NaNs with other payloads)
        }
        printf("%-80s %7zu pairs: %zu with a zero operand -> legacy gives 0 in %zu; %zu without -> same bits as v_mul_f32 in %zu (same up to the NaN's payload: %zu)\n",
               c.name, c.end - c.begin, zero_ops, zero_rule_ok, other, other_same, other_same_mod_nan);
        total_bad += (int)(zero_ops - zero_rule_ok) + (int)(other - other_same_mod_nan);
    }
    printf("%s\n", total_bad ? "MISMATCHES" : "OK: a zero operand wins; every other product is v_mul_f32's");
    return total_bad ? 1 : 0;
}
