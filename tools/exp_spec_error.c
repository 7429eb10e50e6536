// Exhaustive: the relative error of min(exp_spec(x), 1) (csrc/vrt_spec.h, the oracle's vo_expf) against exp(x) in double over EVERY
// fp32 x in [-87, -0] (1.12e9 values, ~20 s on 8 threads).  The constant kDenExpSpec of csrc/vrt_denoise_bound.h rests on it:
//   gcc -O2 -ffp-contract=off -fopenmp -o /tmp/exp_spec_error tools/exp_spec_error.c -lm && /tmp/exp_spec_error
//   -> max rel err 8.131927e-08 (= 1.364 eps, eps = 2^-24) at x = -59.9505272
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <omp.h>
static float exp_spec(float x){
    if (x != x) return x;
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) return INFINITY;
    float fx = floorf(x * 1.44269504088896341f + 0.5f);
    x = x - fx * 0.693359375f;
    x = x - fx * -2.12194440e-4f;
    float z = x * x;
    float p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x
                 + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    int n = (int)fx;
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)(n + 127) << 23;
    return p * s.f;
}
int main(){
    // all negative floats from -0 (0x80000000) to -87 (0xC2AE0000)
    uint32_t lo = 0x80000000u, hi = 0xC2AE0000u;
    double worst = 0; uint32_t wbits = 0; double worst_abs_tiny=0;
    #pragma omp parallel
    {
        double w = 0; uint32_t wb = 0;
        #pragma omp for schedule(static)
        for (int64_t b = lo; b <= (int64_t)hi; b++) {
            union { uint32_t u; float f; } v; v.u = (uint32_t)b;
            float r = exp_spec(v.f);
            if (r > 1.0f) r = 1.0f;           // min(., 1)
            double t = exp((double)v.f);
            double e = fabs((double)r - t) / t;
            if (e > w) { w = e; wb = v.u; }
        }
        #pragma omp critical
        if (w > worst) { worst = w; wbits = wb; }
    }
    union { uint32_t u; float f; } v; v.u = wbits;
    printf("max rel err of min(exp_spec(x),1) on [-87,-0]: %.6e (= %.3f eps, eps=2^-24) at x=%.9g\n", worst, worst/5.9604644775390625e-8, v.f);
    return 0;
}
