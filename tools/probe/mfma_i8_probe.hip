// Microbenchmark (tuning aid, not part of the library): mfma_probe.hip with v_mfma_i32_16x16x64_i8 in place of
// v_mfma_f32_16x16x32_bf16 (same operand bytes per instruction, twice the MACs): is an int8 shadow worth building?
// Random int8 operands.  Prints cycles per MFMA (s_memtime) and the wall-clock rate in POP/s.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

// 32 MFMAs per iteration on a[0:127] (the scan's accumulator file), A/B operands in VGPRs
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD, WAVES_PER_SIMD) void probe_agpr(const u32x4* in, unsigned long long* out,
                                                                                   int iters) {
    u32x4 a0 = in[threadIdx.x], a1 = in[threadIdx.x + 1024], b0 = in[threadIdx.x + 2048], b1 = in[threadIdx.x + 3072];
    unsigned long long t0, t1;
    asm volatile(
        "s_memtime %[t0]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        ".Lloop%=:\n\t"
#define M(i) "v_mfma_i32_16x16x64_i8 a[" #i "*4:" #i "*4+3], %[a0], %[b0], a[" #i "*4:" #i "*4+3]\n\t"
        M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#undef M
#define M(i) "v_mfma_i32_16x16x64_i8 a[" #i "*4:" #i "*4+3], %[a1], %[b1], a[" #i "*4:" #i "*4+3]\n\t"
        M(16) M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)
#undef M
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 .Lloop%=\n\t"
        "s_nop 15\n\t"
        "s_memtime %[t1]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [t0] "=&s"(t0), [t1] "=&s"(t1), [n] "+s"(iters)
        : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1)
        : "memory", "scc",
          "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19",
          "a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39",
          "a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59",
          "a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79",
          "a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99",
          "a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115",
          "a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// the same with the 32 accumulators in VGPRs (compiler-allocated, tied)
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD, WAVES_PER_SIMD) void probe_vgpr(const u32x4* in, unsigned long long* out,
                                                                                   float* sink, int iters) {
    u32x4 a0 = in[threadIdx.x], a1 = in[threadIdx.x + 1024], b0 = in[threadIdx.x + 2048], b1 = in[threadIdx.x + 3072];
    f32x4 c[32];
    for (int i = 0; i < 32; ++i) c[i] = (f32x4){0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a0), "v"(b0));
#pragma unroll
        for (int i = 16; i < 32; ++i) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a1), "v"(b1));
    }
    asm volatile("s_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    int s = 0;
    for (int i = 0; i < 32; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (s == 12345678) sink[threadIdx.x] = (float)s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const int iters = 20000;
    std::vector<unsigned> h(4096 * 4);
    srand(7);
    for (auto& v : h) v = ((unsigned)rand() << 16) ^ (unsigned)rand();  // random int8 quadruples
    u32x4* din;
    unsigned long long* dout;
    float* sink;
    CK(hipMalloc(&din, h.size() * 4));
    CK(hipMalloc(&dout, 256 * 8 * 8));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto report = [&](const char* name, int waves_per_simd, float ms) {
        std::vector<unsigned long long> t(256 * 4 * waves_per_simd);
        (void)hipMemcpy(t.data(), dout, t.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : t) sum += (double)v;
        const double cyc = sum / t.size();
        const double mfma_per_simd = (double)iters * 32 * waves_per_simd;
        printf("%-28s %d wave(s)/SIMD: %.2f cycles per MFMA per SIMD, wall %.3f ms, %.2f POP/s, clock %.2f GHz\n", name,
               waves_per_simd, cyc / mfma_per_simd, ms, 1024.0 * mfma_per_simd * 32768 / (ms * 1e-3) / 1e15, cyc / (ms * 1e-3) / 1e9);
    };
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        CK(hipEventRecord(e0));
        probe_agpr<1><<<256, 256>>>(din, dout, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        report("AGPR accumulators", 1, ms);
        CK(hipEventRecord(e0));
        probe_agpr<2><<<256, 512>>>(din, dout, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        report("AGPR accumulators", 2, ms);
        CK(hipEventRecord(e0));
        probe_vgpr<1><<<256, 256>>>(din, dout, sink, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        report("VGPR accumulators", 1, ms);
        CK(hipEventRecord(e0));
        probe_vgpr<2><<<256, 512>>>(din, dout, sink, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        report("VGPR accumulators", 2, ms);
    }
    return 0;
}
