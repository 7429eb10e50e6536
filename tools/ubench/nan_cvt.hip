// v_cvt_rpi_i32_f32 of NaN / inf / large values on gfx950 (the K1 fast loop relies on NaN -> 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* in, int* out, int n)
{
    int i = threadIdx.x;
    if (i >= n) return;
    float v = in[i]; int r;
    asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    out[i] = r;
    float inf = INFINITY, zero = 0.0f, c = 0.0f, f;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f) : "v"(inf), "v"(zero), "v"(c));
    asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    out[n + i] = r;
}
int main()
{
    float h[8] = {NAN, -NAN, INFINITY, -INFINITY, 2.5f, -2.5f, 3.4999f, -1023.49f};
    float* d; int* o; hipMalloc(&d, 32); hipMalloc(&o, 64);
    hipMemcpy(d, h, 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 8);
    int r[16]; hipMemcpy(r, o, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; i++) printf("cvt_rpi(%g) = %d ; cvt_rpi(fma(inf,0,0)) = %d\n", h[i], r[i], r[8 + i]);
    return 0;
}
