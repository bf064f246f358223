// Microbenchmark (tuning aid, not part of the library): the scan's inner instruction mix -- one ds_read_b128 of a B
// fragment per 65,536 int8 operations (32,768 multiply-adds), two waves per SIMD -- with v_mfma_i32_16x16x64_i8 (two MFMAs per fragment, what
// the scan body issues) against v_mfma_i32_32x32x32_i8 (one MFMA per fragment: half the MFMA issue slots for the same
// work).  Random int8 operands, accumulators a[0:127] in both forms.  Prints cycles per fragment and the wall-clock rate.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CLOB                                                                                                              \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18",  \
        "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35",   \
        "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52",   \
        "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69",   \
        "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86",   \
        "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102",       \
        "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117",  \
        "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"

// SHAPE 0: 32 fragments x 2 v_mfma_i32_16x16x64_i8 (accumulators of 4 registers); SHAPE 1: 32 fragments x 1
// v_mfma_i32_32x32x32_i8 (accumulators of 16 registers, 8 of them, each updated 4 times per iteration)
template <int SHAPE>
__global__ __launch_bounds__(512, 2) void probe(const u32x4* in, unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem);
    for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = in[i];  // 32 KiB of B fragments
    __syncthreads();
    u32x4 a0 = in[threadIdx.x + 4096], a1 = in[threadIdx.x + 5120];
    u32x4 t0r, t1r, t2r, t3r;
    const unsigned ldr = (threadIdx.x & 63) * 16;
    unsigned long long t0, t1;
    if (SHAPE == 0) {
        asm volatile(
            "s_memtime %[t0]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "ds_read_b128 %[q0], %[ldr]\n\t"
            "ds_read_b128 %[q1], %[ldr] offset:1024\n\t"
            "ds_read_b128 %[q2], %[ldr] offset:2048\n\t"
            "ds_read_b128 %[q3], %[ldr] offset:3072\n\t"
            ".Lloop%=:\n\t"
#define F(i, q, off)                                                                                              \
    "s_waitcnt lgkmcnt(3)\n\t"                                                                                    \
    "v_mfma_i32_16x16x64_i8 a[" #i "*8:" #i "*8+3], %[a0], %[" #q "], a[" #i "*8:" #i "*8+3]\n\t"                  \
    "v_mfma_i32_16x16x64_i8 a[" #i "*8+4:" #i "*8+7], %[a1], %[" #q "], a[" #i "*8+4:" #i "*8+7]\n\t"              \
    "ds_read_b128 %[" #q "], %[ldr] offset:" #off "\n\t"
            F(0, q0, 4096) F(1, q1, 5120) F(2, q2, 6144) F(3, q3, 7168) F(4, q0, 8192) F(5, q1, 9216) F(6, q2, 10240)
            F(7, q3, 11264) F(8, q0, 12288) F(9, q1, 13312) F(10, q2, 14336) F(11, q3, 15360) F(12, q0, 16384)
            F(13, q1, 17408) F(14, q2, 18432) F(15, q3, 19456)
#undef F
#define F(i, q, off)                                                                                              \
    "s_waitcnt lgkmcnt(3)\n\t"                                                                                    \
    "v_mfma_i32_16x16x64_i8 a[" #i "*8:" #i "*8+3], %[a1], %[" #q "], a[" #i "*8:" #i "*8+3]\n\t"                  \
    "v_mfma_i32_16x16x64_i8 a[" #i "*8+4:" #i "*8+7], %[a0], %[" #q "], a[" #i "*8+4:" #i "*8+7]\n\t"              \
    "ds_read_b128 %[" #q "], %[ldr] offset:" #off "\n\t"
            F(0, q0, 20480) F(1, q1, 21504) F(2, q2, 22528) F(3, q3, 23552) F(4, q0, 24576) F(5, q1, 25600) F(6, q2, 26624)
            F(7, q3, 27648) F(8, q0, 28672) F(9, q1, 29696) F(10, q2, 30720) F(11, q3, 31744) F(12, q0, 0) F(13, q1, 1024)
            F(14, q2, 2048) F(15, q3, 3072)
#undef F
            "s_sub_u32 %[n], %[n], 1\n\t"
            "s_cmp_lg_u32 %[n], 0\n\t"
            "s_cbranch_scc1 .Lloop%=\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_nop 15\n\t"
            "s_memtime %[t1]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            : [t0] "=&s"(t0), [t1] "=&s"(t1), [n] "+s"(iters), [q0] "=&v"(t0r), [q1] "=&v"(t1r), [q2] "=&v"(t2r), [q3] "=&v"(t3r)
            : [a0] "v"(a0), [a1] "v"(a1), [ldr] "v"(ldr)
            : "memory", "scc", CLOB);
    } else {
        asm volatile(
            "s_memtime %[t0]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "ds_read_b128 %[q0], %[ldr]\n\t"
            "ds_read_b128 %[q1], %[ldr] offset:1024\n\t"
            "ds_read_b128 %[q2], %[ldr] offset:2048\n\t"
            "ds_read_b128 %[q3], %[ldr] offset:3072\n\t"
            ".Lloop%=:\n\t"
#define F(i, q, off, av)                                                                                          \
    "s_waitcnt lgkmcnt(3)\n\t"                                                                                    \
    "v_mfma_i32_32x32x32_i8 a[" #i "*16:" #i "*16+15], %[" #av "], %[" #q "], a[" #i "*16:" #i "*16+15]\n\t"        \
    "ds_read_b128 %[" #q "], %[ldr] offset:" #off "\n\t"
            F(0, q0, 4096, a0) F(1, q1, 5120, a0) F(2, q2, 6144, a0) F(3, q3, 7168, a0) F(4, q0, 8192, a0) F(5, q1, 9216, a0)
            F(6, q2, 10240, a0) F(7, q3, 11264, a0) F(0, q0, 12288, a1) F(1, q1, 13312, a1) F(2, q2, 14336, a1)
            F(3, q3, 15360, a1) F(4, q0, 16384, a1) F(5, q1, 17408, a1) F(6, q2, 18432, a1) F(7, q3, 19456, a1)
            F(0, q0, 20480, a0) F(1, q1, 21504, a0) F(2, q2, 22528, a0) F(3, q3, 23552, a0) F(4, q0, 24576, a0)
            F(5, q1, 25600, a0) F(6, q2, 26624, a0) F(7, q3, 27648, a0) F(0, q0, 28672, a1) F(1, q1, 29696, a1)
            F(2, q2, 30720, a1) F(3, q3, 31744, a1) F(4, q0, 0, a1) F(5, q1, 1024, a1) F(6, q2, 2048, a1) F(7, q3, 3072, a1)
#undef F
            "s_sub_u32 %[n], %[n], 1\n\t"
            "s_cmp_lg_u32 %[n], 0\n\t"
            "s_cbranch_scc1 .Lloop%=\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_nop 15\n\t"
            "s_memtime %[t1]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            : [t0] "=&s"(t0), [t1] "=&s"(t1), [n] "+s"(iters), [q0] "=&v"(t0r), [q1] "=&v"(t1r), [q2] "=&v"(t2r), [q3] "=&v"(t3r)
            : [a0] "v"(a0), [a1] "v"(a1), [ldr] "v"(ldr)
            : "memory", "scc", CLOB);
    }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const int iters = 20000;
    std::vector<unsigned> h(8192 * 4);
    srand(7);
    for (auto& v : h) v = ((unsigned)rand() << 16) ^ (unsigned)rand();
    u32x4* din;
    unsigned long long* dout;
    CK(hipMalloc(&din, h.size() * 4));
    CK(hipMalloc(&dout, 256 * 8 * 8));
    CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep)
        for (int shape = 0; shape < 2; ++shape) {
            float ms;
            CK(hipEventRecord(e0));
            if (shape == 0) probe<0><<<256, 512, 32768>>>(din, dout, iters);
            else probe<1><<<256, 512, 32768>>>(din, dout, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> t(256 * 8);
            CK(hipMemcpy(t.data(), dout, t.size() * 8, hipMemcpyDeviceToHost));
            double sum = 0;
            for (auto v : t) sum += (double)v;
            const double cyc = sum / t.size();
            const double frags_per_simd = (double)iters * 32 * 2;  // two waves per SIMD
            printf("%-34s %.2f cycles per fragment (65536 int8 ops) per SIMD, wall %.3f ms, %.2f POP/s, clock %.2f GHz\n",
                   shape == 0 ? "2 x v_mfma_i32_16x16x64_i8 + 1 read" : "1 x v_mfma_i32_32x32x32_i8 + 1 read",
                   cyc / frags_per_simd, ms, 1024.0 * frags_per_simd * 65536 / (ms * 1e-3) / 1e15, cyc / (ms * 1e-3) / 1e9);
        }
    return 0;
}
