// hwmath_accuracy.hip -- error of the gfx950 hardware transcendentals the FAST kernels use, against binary64.
//   hipcc --offload-arch=gfx950 -O2 tools/hwmath_accuracy.hip -o /tmp/hwmath && /tmp/hwmath
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void run(int fn, int n, const float* x, float* y)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float a = x[i], r = 0;
    switch (fn)
    {
    case 0: r = __builtin_amdgcn_sinf(a); break;  // sin(2 pi a)
    case 1: r = __builtin_amdgcn_cosf(a); break;  // cos(2 pi a)
    case 2: r = __builtin_amdgcn_rcpf(a); break;
    case 3: r = __builtin_amdgcn_rsqf(a); break;
    case 4: r = __builtin_amdgcn_sqrtf(a); break;
    case 5: r = __builtin_amdgcn_logf(a); break;  // log2
    case 6: r = __builtin_amdgcn_exp2f(a); break;
    }
    y[i] = r;
}

int main()
{
    const int n = 1 << 20;
    const char* names[] = {"v_sin_f32 (rev)", "v_cos_f32 (rev)", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_log_f32", "v_exp_f32"};
    std::vector<float> x(n), y(n);
    float *dx, *dy;
    hipMalloc(&dx, n * 4);
    hipMalloc(&dy, n * 4);
    for (int fn = 0; fn < 7; fn++)
    {
        srand(1);
        for (int i = 0; i < n; i++)
        {
            double u = (rand() + 0.5) / (RAND_MAX + 1.0);
            x[i] = fn < 2 ? (float)u : fn == 6 ? (float)(-20.0 * u) : (float)(u * 4.0 + 1e-3);
        }
        hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
        run<<<n / 256, 256>>>(fn, n, dx, dy);
        hipMemcpy(y.data(), dy, n * 4, hipMemcpyDeviceToHost);
        double maxAbs = 0, maxUlp = 0, sumUlp = 0;
        for (int i = 0; i < n; i++)
        {
            double a = x[i], w;
            switch (fn)
            {
            case 0: w = sin(2 * M_PI * a); break;
            case 1: w = cos(2 * M_PI * a); break;
            case 2: w = 1 / a; break;
            case 3: w = 1 / sqrt(a); break;
            case 4: w = sqrt(a); break;
            case 5: w = log2(a); break;
            default: w = exp2(a); break;
            }
            double e = fabs(y[i] - w);
            double ulp = e / (fabs(w) > 1e-30 ? ldexp(1.0, ilogb(w) - 23) : 1e-45);
            if (e > maxAbs) maxAbs = e;
            if (ulp > maxUlp) maxUlp = ulp;
            sumUlp += ulp;
        }
        printf("%-18s max abs err %.3g   max ulp %.3g   mean ulp %.3g\n", names[fn], maxAbs, maxUlp, sumUlp / n);
    }
    return 0;
}
