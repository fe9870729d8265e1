#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* y0, double* y1, double* y2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rsq(v);
    y0[i] = r * r * r;
    double h = 0.5 * v;
    double a = r * (1.5 - h * r * r);
    y1[i] = a * a * a;
    double b = a * (1.5 - h * a * a);
    y2[i] = b * b * b;
}
int main()
{
    const int n = 1 << 20;
    double *hx = new double[n], *h0 = new double[n], *h1 = new double[n], *h2 = new double[n];
    for (int i = 0; i < n; i++) hx[i] = 0.04 + 3.0 * (double)rand() / RAND_MAX;  // x = 1+g^2-2g cos in (0.04, 3)
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) {
        double ref = pow(hx[i], -1.5);
        e0 = fmax(e0, fabs(h0[i] - ref) / ref); e1 = fmax(e1, fabs(h1[i] - ref) / ref); e2 = fmax(e2, fabs(h2[i] - ref) / ref);
    }
    printf("max rel err x^-1.5: raw rsq %.3e, 1 Newton %.3e, 2 Newton %.3e\n", e0, e1, e2);
    return 0;
}
