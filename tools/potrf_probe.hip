// Developer probe (not part of the library): the 32 x 32 diagonal factor of the blocked Cholesky
// (csrc/chol_diag.h) in its two forms on one wave - time per block, residual |L L^T - A|, difference
// between the forms - and the accuracy of v_rsq_f64, which decides how many correction steps the
// reciprocal square root needs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zuds-pipeline_amd/csrc tools/potrf_probe.hip -o tools/_build/potrf_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "chol_diag.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_rsq(const double* x, double* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __builtin_amdgcn_rsq(x[i]);
}

// REP factorisations of the same block by one wave; the block is restored from a copy in LDS each time
template <int FORM>
__global__ __launch_bounds__(64) void k_potrf(const double* A, double* L, int rep, long long* ticks, int* fail) {
    __shared__ double D[CH_NB][CH_NB + 1];
    __shared__ double A0[CH_NB][CH_NB + 1];
    const int lane = threadIdx.x;
    for (int e = lane; e < CH_NB * CH_NB; e += 64) A0[e >> 5][e & 31] = A[e];
    __syncthreads();
    long long t = 0, tc = 0;
    for (int r = 0; r < rep; ++r) {
        for (int e = lane; e < CH_NB * CH_NB; e += 64) D[e >> 5][e & 31] = A0[e >> 5][e & 31];
        __syncthreads();
        const long long t0 = wall_clock64(), c0 = clock64();
        if (FORM == 0) chol_diag_wave_panel_ref<0>(D, CH_NB, fail);
        else if (FORM == 1) chol_diag_wave_panel_fast<0>(D, CH_NB, fail);
        else chol_diag_wave_panel_dpp<0>(D, CH_NB, fail);
        __syncthreads();
        t += wall_clock64() - t0;
        tc += clock64() - c0;
    }
    for (int e = lane; e < CH_NB * (CH_NB + 1); e += 64) L[e] = D[e / (CH_NB + 1)][e % (CH_NB + 1)];
    if (lane == 0) { ticks[0] = t; ticks[1] = tc; }
}

int main() {
    // ---- v_rsq_f64 against 1 / sqrt in long double
    {
        const int n = 1 << 20;
        std::vector<double> x(n), y(n);
        srand(5);
        for (int i = 0; i < n; ++i) {
            const double m = 1.0 + (double)rand() / RAND_MAX * 3.0;         // mantissas over [1, 4): both exponent parities
            x[i] = ldexp(m, (rand() % 61) - 30);
        }
        double *dx, *dy;
        CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dy, n * 8));
        CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_rsq, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
        CK(hipMemcpy(y.data(), dy, n * 8, hipMemcpyDeviceToHost));
        long double worst = 0, sum = 0;
        for (int i = 0; i < n; ++i) {
            const long double ref = 1.0L / sqrtl((long double)x[i]);
            const long double rel = fabsl(((long double)y[i] - ref) / ref);
            worst = rel > worst ? rel : worst;
            sum += rel;
        }
        printf("v_rsq_f64: worst relative error %.3Le = 2^%.1Lf, mean %.3Le (n = %d)\n", worst, log2l(worst), sum / n, n);
        printf("  one second-order step would stop at 1.5 e^2 = 2^%.1Lf, the third-order step at 2.5 e^3 = 2^%.1Lf\n",
               log2l(1.5L * worst * worst), log2l(2.5L * worst * worst * worst));
    }
    // ---- the factor: an SPD block of the kind the Jacobi-scaled normal matrix has (unit diagonal, correlated)
    std::vector<double> A(CH_NB * CH_NB);
    srand(11);
    {
        std::vector<double> B(CH_NB * 48);
        for (auto& v : B) v = (double)rand() / RAND_MAX - 0.3;
        for (int i = 0; i < CH_NB; ++i)
            for (int j = 0; j < CH_NB; ++j) {
                double s = 0;
                for (int k = 0; k < 48; ++k) s += B[i * 48 + k] * B[j * 48 + k];
                A[i * CH_NB + j] = s;
            }
        std::vector<double> d(CH_NB);
        for (int i = 0; i < CH_NB; ++i) d[i] = sqrt(A[i * CH_NB + i]);
        for (int i = 0; i < CH_NB; ++i)
            for (int j = 0; j < CH_NB; ++j) A[i * CH_NB + j] /= d[i] * d[j];
        for (int i = 0; i < CH_NB; ++i) A[i * CH_NB + i] += 1e-10;
    }
    double *dA, *dL;
    long long* dt;
    int* dfail;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dL, CH_NB * (CH_NB + 1) * 8)); CK(hipMalloc(&dt, 16)); CK(hipMalloc(&dfail, 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> L[3];
    for (int form = 0; form < 3; ++form) {
        L[form].resize(CH_NB * (CH_NB + 1));
        const int rep = 2000;
        long long ticks = 0, tk[2] = {0, 0};
        double best = 1e30, bestc = 0;
        for (int pass = 0; pass < 5; ++pass) {
            CK(hipMemset(dfail, 0, 4));
            if (form == 0) hipLaunchKernelGGL(k_potrf<0>, dim3(1), dim3(64), 0, 0, dA, dL, rep, dt, dfail);
            else if (form == 1) hipLaunchKernelGGL(k_potrf<1>, dim3(1), dim3(64), 0, 0, dA, dL, rep, dt, dfail);
            else hipLaunchKernelGGL(k_potrf<2>, dim3(1), dim3(64), 0, 0, dA, dL, rep, dt, dfail);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(tk, dt, 16, hipMemcpyDeviceToHost));
            printf("  form %d pass %d: %.3f us, %.0f shader clocks per block\n", form, pass, tk[0] * 0.01 / rep, (double)tk[1] / rep);
            if (tk[0] * 0.01 / rep < best) { best = tk[0] * 0.01 / rep; bestc = (double)tk[1] / rep; }
            ticks = tk[0];
        }
        (void)bestc;
        int fail = 0;
        CK(hipMemcpy(&fail, dfail, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(L[form].data(), dL, L[form].size() * 8, hipMemcpyDeviceToHost));
        // residual and the reciprocal diagonal
        long double res = 0, rinv = 0;
        for (int i = 0; i < CH_NB; ++i) {
            for (int j = 0; j <= i; ++j) {
                long double s = 0;
                for (int k = 0; k <= j; ++k) s += (long double)L[form][i * (CH_NB + 1) + k] * L[form][j * (CH_NB + 1) + k];
                const long double r = fabsl(s - A[i * CH_NB + j]);
                res = r > res ? r : res;
            }
            const long double q = fabsl((long double)L[form][i * (CH_NB + 1) + CH_NB] * L[form][i * (CH_NB + 1) + i] - 1.0L);
            rinv = q > rinv ? q : rinv;
        }
        printf("form %d (%s): %.2f us per 32 x 32 block (wall clock 100 MHz, %d repetitions), max |L L^T - A| %.2Le, "
               "max |dinv L_ii - 1| %.2Le, clamped pivots %d\n", form, form == 0 ? "rounds 1 - 3" : form == 1 ? "round 4" : "round 5: DPP broadcasts",
               ticks * 0.01 / rep, rep, res, rinv, fail / rep);
    }
    double dmax = 0;
    for (int i = 0; i < CH_NB; ++i)
        for (int j = 0; j <= i; ++j) dmax = fmax(dmax, fabs(L[0][i * (CH_NB + 1) + j] - L[1][i * (CH_NB + 1) + j]));
    printf("max |L(round 4) - L(rounds 1 - 3)| = %.2e\n", dmax);
    {
        // the DPP form against the round-4 form: the same operations per entry, so the same bits (lower triangle and
        // the reciprocal diagonal in column 32)
        int ndiff = 0;
        for (int i = 0; i < CH_NB; ++i) {
            for (int j = 0; j <= i; ++j)
                if (memcmp(&L[2][i * (CH_NB + 1) + j], &L[1][i * (CH_NB + 1) + j], 8)) ++ndiff;
            if (memcmp(&L[2][i * (CH_NB + 1) + CH_NB], &L[1][i * (CH_NB + 1) + CH_NB], 8)) ++ndiff;
        }
        printf("entries of L(round 5, DPP) that differ in any bit from L(round 4): %d\n", ndiff);
        if (ndiff) return 1;
    }
    return 0;
}
