// Issue cost of v_pk_mul_f32 / v_pk_add_f32 against v_mul_f32 / v_add_f32 on gfx950, 1 and 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int N = 4096;   // iterations, each 8 independent chains x 2 ops
__global__ void k_scalar(float *out, float a, float b) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    for (int it = 0; it < N; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_packed(float *out, float a, float b) {
    f2 x[8]; f2 av = {a, a}, bv = {b, b};
    for (int i = 0; i < 8; ++i) x[i] = f2{(float)threadIdx.x + i, (float)threadIdx.x - i};
    for (int it = 0; it < N; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(av));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(bv));
    }
    f2 s = {0, 0}; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
// dependent chain (one accumulator): latency rather than issue
__global__ void k_scalar_dep(float *out, float a, float b) {
    float x = threadIdx.x, y = threadIdx.x + 1;
    for (int it = 0; it < N * 8; ++it) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(y) : "v"(a)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(y) : "v"(b)); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
__global__ void k_packed_dep(float *out, float a, float b) {
    f2 x = {(float)threadIdx.x, (float)threadIdx.x + 1}; f2 av = {a, a}, bv = {b, b};
    for (int it = 0; it < N * 8; ++it) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(av)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(bv)); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x.x + x.y;
}
template <typename K> float run(K k, int blocks, int threads, float *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, threads>>>(d, 1.0001f, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<blocks, threads>>>(d, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float *d; hipMalloc(&d, 1 << 24);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * 4 * wps;   // one 64-lane workgroup per SIMD slot
        float a = run(k_scalar, blocks, 64, d), b = run(k_packed, blocks, 64, d);
        float c = run(k_scalar_dep, blocks, 64, d), e = run(k_packed_dep, blocks, 64, d);
        // per wave: N*32 scalar instrs vs N*16 packed instrs for the same flops
        printf("waves/SIMD %d: scalar %.3f ms (%.2f ns/instr/wave) packed %.3f ms (%.2f ns/instr/wave)  same work ratio packed/scalar %.2f | dep scalar %.3f packed %.3f ratio %.2f\n", wps,
               a, a * 1e6 / (N * 32), b, b * 1e6 / (N * 16), b / a, c, e, e / c);
    }
    return 0;
}
