// v_mul_legacy_f32 on gfx950: 0 x anything = 0 (DX9 rule)?  Everything else as v_mul_f32?   hipcc --offload-arch=gfx950 -O2 mul_legacy.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const float* a, const float* b, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a[i]), "v"(b[i]));
    o[i] = r;
}
int main() {
    const float nan = std::nanf(""), inf = INFINITY;
    std::vector<float> a = {0.f, 0.f, 0.f, -0.f, 1.f, 2.f, nan, 0.f, 1e-30f, 3.f}, b = {nan, inf, -inf, nan, nan, 3.5f, 0.f, 5.f, 1e-30f, inf};
    // + random pairs: must equal the ordinary product bit for bit
    unsigned s = 12345;
    for (int i = 0; i < 100000; ++i) {
        s = s * 1664525u + 1013904223u; float x; unsigned u = (s >> 9) | 0x3f800000u; memcpy(&x, &u, 4);
        s = s * 1664525u + 1013904223u; float y; u = (s >> 9) | 0x3f800000u; memcpy(&y, &u, 4);
        a.push_back((x - 1.5f) * 1e3f); b.push_back((y - 1.5f) * 1e-3f);
    }
    const int n = (int)a.size();
    float *da, *db, *dout; hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dout, n);
    std::vector<float> o(n); hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 10; ++i) printf("%g x %g = %g\n", a[i], b[i], o[i]);
    int diff = 0;
    for (int i = 10; i < n; ++i) { float w = a[i] * b[i]; if (memcmp(&w, &o[i], 4)) ++diff; }
    printf("random pairs differing from the ordinary product: %d of %d\n", diff, n - 10);
    return 0;
}
