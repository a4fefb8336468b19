// v_pk_mov_b32 operand selection on gfx950, determined empirically: prints, for each
// (op_sel, op_sel_hi) combination, which halves of src0 = (1,2) and src1 = (3,4) land in dst.
//   hipcc --offload-arch=gfx950 -O2 -o pk_mov tools/ubench/pk_mov.hip && ./pk_mov
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CASE(i, SEL, SELHI)                                                                                  \
    {                                                                                                        \
        v2f d;                                                                                               \
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:" SEL " op_sel_hi:" SELHI : "=v"(d) : "v"(a), "v"(b)); \
        out[2 * i] = d.x;                                                                                    \
        out[2 * i + 1] = d.y;                                                                                \
    }

__global__ void k(float* out) {
    v2f a = {1.f, 2.f}, b = {3.f, 4.f};
    CASE(0, "[0,0]", "[0,0]") CASE(1, "[0,0]", "[0,1]") CASE(2, "[0,0]", "[1,0]") CASE(3, "[0,0]", "[1,1]")
    CASE(4, "[0,1]", "[0,0]") CASE(5, "[0,1]", "[0,1]") CASE(6, "[0,1]", "[1,0]") CASE(7, "[0,1]", "[1,1]")
    CASE(8, "[1,0]", "[0,0]") CASE(9, "[1,0]", "[0,1]") CASE(10, "[1,0]", "[1,0]") CASE(11, "[1,0]", "[1,1]")
    CASE(12, "[1,1]", "[0,0]") CASE(13, "[1,1]", "[0,1]") CASE(14, "[1,1]", "[1,0]") CASE(15, "[1,1]", "[1,1]")
}

int main() {
    float* d;
    float h[32];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* sel[4] = {"[0,0]", "[0,1]", "[1,0]", "[1,1]"};
    for (int i = 0; i < 16; ++i) printf("op_sel:%s op_sel_hi:%s -> (%g, %g)\n", sel[i / 4], sel[i % 4], h[2 * i], h[2 * i + 1]);
    return 0;
}
