// Development check: is  q1 = fma(fma(-q0, s, x), rs, q0),  q0 = x * rs,  rs = RN(1/s)
// bit-identical to the IEEE fp32 quotient x / s?  (Markstein's correction with a correctly rounded reciprocal.)
// Counts mismatches over (a) random operand pairs of several magnitude classes and (b) every significand of x for
// a set of divisors, and reports the divisors (significand patterns) involved in any mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t rng(uint64_t& st) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    return (uint32_t)(st >> 16);
}

__device__ __forceinline__ float fast_div(float x, float s, float rs) {
    const float q0 = x * rs;
    const float r = __builtin_fmaf(-q0, s, x);
    return __builtin_fmaf(r, rs, q0);
}

// mode 0: x, s random significands, exponent of x in [-100, 100], of s in [-40, 40]; only quotients inside the guarded range of
// Divisor<float> (2^-80 <= |x * rs| <= 2^100) are compared
// mode 1: s from a table of nasty significands, x exhaustive over the 2^23 significands at exponent ex
__global__ void k_random(uint64_t seed, unsigned long long* bad, unsigned long long* bad_allones, float* sample) {
    uint64_t st = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    for (int i = 0; i < 4; ++i) rng(st);
    unsigned long long nb = 0, nb1 = 0;
    for (int it = 0; it < 4096; ++it) {
        const uint32_t mx = rng(st) & 0x7fffffu, ms = rng(st) & 0x7fffffu, ee = rng(st);
        int ex = (int)(ee & 255) - 128, es = (int)((ee >> 8) & 127) - 64;
        ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
        es = es < -40 ? -40 : (es > 40 ? 40 : es);
        const uint32_t sx = (ee >> 20) & 1u;
        float x = __uint_as_float((sx << 31) | ((uint32_t)(127 + ex) << 23) | mx);
        float s = __uint_as_float(((uint32_t)(127 + es) << 23) | ms);
        const float rs = 1.0f / s;
        const float a = x / s, b = fast_div(x, s, rs);
        const float q0a = fabsf(x * rs);
        const bool guarded = q0a <= 0x1p100f && q0a >= 0x1p-80f;  // outside: Divisor<float> takes the IEEE path
        if (guarded && __float_as_uint(a) != __float_as_uint(b)) {
            if (ms == 0x7fffffu) ++nb1; else { ++nb; sample[0] = x; sample[1] = s; }
        }
    }
    if (nb) atomicAdd(bad, nb);
    if (nb1) atomicAdd(bad_allones, nb1);
}

__global__ void k_exhaust(const uint32_t* sig, int nsig, int ex, int es, unsigned long long* bad, float* sample) {
    const uint32_t mx = blockIdx.x * blockDim.x + threadIdx.x;  // 2^23 threads
    const float x = __uint_as_float(((uint32_t)(127 + ex) << 23) | mx);
    unsigned long long nb = 0;
    for (int j = 0; j < nsig; ++j) {
        const float s = __uint_as_float(((uint32_t)(127 + es) << 23) | sig[j]);
        const float rs = 1.0f / s;
        if (__float_as_uint(x / s) != __float_as_uint(fast_div(x, s, rs))) { ++nb; sample[0] = x; sample[1] = s; }
    }
    if (nb) atomicAdd(bad, nb);
}

int main() {
    unsigned long long *bad, *bad1;
    float* sample;
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&bad1, 8)); CK(hipMalloc(&sample, 8));
    CK(hipMemset(bad, 0, 8)); CK(hipMemset(bad1, 0, 8)); CK(hipMemset(sample, 0, 8));
    const int blocks = 4096, threads = 256, rounds = 32;
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_random, dim3(blocks), dim3(threads), 0, 0, 1234567ull + r * 7919ull, bad, bad1, sample);
    CK(hipDeviceSynchronize());
    unsigned long long hb, hb1; float hs[2];
    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb1, bad1, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, sample, 8, hipMemcpyDeviceToHost));
    printf("random pairs: %llu   mismatches (divisor significand not all ones): %llu   (all ones): %llu   last bad x=%a s=%a\n",
           (unsigned long long)blocks * threads * rounds * 4096ull, hb, hb1, hs[0], hs[1]);
    // exhaustive over x significands for special divisors
    uint32_t h_sig[] = {0x000000u, 0x000001u, 0x000002u, 0x000003u, 0x7ffffeu, 0x7ffffdu, 0x7ffffcu, 0x400000u, 0x400001u, 0x3fffffu,
                        0x2aaaaau, 0x2aaaabu, 0x555555u, 0x555556u, 0x199999u, 0x19999au, 0x666666u, 0x4ccccdu, 0x123456u, 0x7f0000u,
                        0x00ffffu, 0x100000u, 0x200000u, 0x600000u, 0x700000u, 0x780000u, 0x7c0000u, 0x7e0000u, 0x0000ffu, 0x00000fu,
                        0x7ffff0u, 0x7fff00u};
    const int nsig = sizeof(h_sig) / 4;
    uint32_t* d_sig; CK(hipMalloc(&d_sig, sizeof(h_sig))); CK(hipMemcpy(d_sig, h_sig, sizeof(h_sig), hipMemcpyHostToDevice));
    CK(hipMemset(bad, 0, 8));
    for (int ex = -3; ex <= 3; ++ex)
        for (int es = -2; es <= 2; es += 2)
            hipLaunchKernelGGL(k_exhaust, dim3((1u << 23) / 256), dim3(256), 0, 0, d_sig, nsig, ex, es, bad, sample);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, sample, 8, hipMemcpyDeviceToHost));
    printf("exhaustive x significands for %d special divisors x 21 exponent pairs: mismatches %llu   last bad x=%a s=%a\n", nsig, hb, hs[0], hs[1]);
    return 0;
}
