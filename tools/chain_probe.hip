// Calibration: how fast can ONE lane run a dependent chain of f64 adds on gfx950?  (a) in registers, (b) through LDS as
// wave_serial_prefix_f64 does (read 64 values, add, write back), (c) the same with values pre-loaded into registers by
// all lanes and broadcast with v_readlane.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void reg_chain(double *out, int n, double x) {
    double r = 0.0;
    if (threadIdx.x == 0) {
        for (int i = 0; i < n; ++i) r = r + x;
        out[blockIdx.x] = r;
    }
}
__global__ void lds_chain(double *out, int chunks, double x) {
    __shared__ double buf[64];
    double carry = 0.0;
    for (int c = 0; c < chunks; ++c) {
        buf[threadIdx.x] = x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (threadIdx.x == 0) {
            double r = carry;
#pragma unroll
            for (int l = 0; l < 64; ++l) {
                r = r + buf[l];
                buf[l] = r;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        carry = buf[63];
        x = x + buf[threadIdx.x] * 1e-300; // keep the per-lane prefix alive
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (threadIdx.x == 0) out[blockIdx.x] = carry + x;
}
// scalar-operand chain: lane values are read with v_readlane into SGPRs; the adds run on lane 0 with an SGPR operand
__global__ void readlane_chain(double *out, int chunks, double x) {
    double carry = 0.0;
    double acc = 0.0;
    for (int c = 0; c < chunks; ++c) {
        double v = x + (double)threadIdx.x * 1e-9;
        const unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)(__double_as_longlong(v) >> 32);
        double r = carry;
#pragma unroll
        for (int l = 0; l < 64; ++l) {
            const unsigned a = __builtin_amdgcn_readlane(lo, l), b = __builtin_amdgcn_readlane(hi, l);
            r = r + __longlong_as_double(((long long)b << 32) | a);
            if ((int)threadIdx.x == l) acc = r;
        }
        carry = r;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

// (d) what hw2_prefix_kernel can do: all 64 values fetched from LDS before the chain starts (register-resident chain),
// results stored by lane 0 straight to global memory
__global__ void lds_upfront_chain(double *out, double *pref, int chunks, double x) {
    __shared__ __align__(16) double buf[64];
    double carry = 0.0;
    for (int c = 0; c < chunks; ++c) {
        buf[threadIdx.x] = x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (threadIdx.x == 0) {
            double t[64];
#pragma unroll
            for (int l = 0; l < 64; ++l) t[l] = buf[l];
            double r = carry;
#pragma unroll
            for (int l = 0; l < 64; ++l) {
                r = r + t[l];
                t[l] = r;
            }
            double *dst = pref + (size_t)(c & 1023) * 64;
#pragma unroll
            for (int l = 0; l < 64; ++l) dst[l] = t[l];
            carry = r;
        }
        carry = __shfl(carry, 0, 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (threadIdx.x == 0) out[blockIdx.x] = carry;
}

int main() {
    double *out;
    hipMalloc(&out, 8 * 64 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms;
    const int n = 1 << 22;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(reg_chain, dim3(1), dim3(64), 0, 0, out, n, 1.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("register chain : %.2f ns per add\n", ms * 1e6 / n);
        hipEventRecord(e0);
        hipLaunchKernelGGL(lds_chain, dim3(1), dim3(64), 0, 0, out, n / 64, 1.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("LDS chain      : %.2f ns per element\n", ms * 1e6 / n);
        hipEventRecord(e0);
        hipLaunchKernelGGL(readlane_chain, dim3(1), dim3(64), 0, 0, out, n / 64, 1.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("readlane chain : %.2f ns per element\n", ms * 1e6 / n);
    }
    double *pref;
    hipMalloc(&pref, 8 * 64 * 1024 * 32);
    for (int wg : {1, 1024}) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(lds_upfront_chain, dim3(wg), dim3(64), 0, 0, out, pref + (size_t)0, n / 64 / (wg > 1 ? 16 : 1), 1.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("LDS upfront + register chain + global stores, %d wavefront(s): %.2f ns per element per wavefront\n", wg,
               ms * 1e6 / (n / (wg > 1 ? 16 : 1)));
    }
    // 256 waves at once (one per CU): does the per-wave rate hold?
    hipEventRecord(e0);
    hipLaunchKernelGGL(lds_chain, dim3(1024), dim3(64), 0, 0, out, n / 64 / 16, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("LDS chain, 1024 wavefronts: %.2f ns per element per wavefront\n", ms * 1e6 / (n / 16));
    return 0;
}
