// Tuning aid (not part of the library): the filter scan's k-loop rebuilt level by level.
//  0 bare MFMAs (one B operand)   1 + rotating A/B operands   2 + s_waitcnt per fragment
//  3 + ds_read_b128 per fragment (4 in flight)   4 + s_barrier per chunk   5 + Q staging writes
//  6 + Q staging loads (L2)
// Prints cycles per MFMA per SIMD (s_memtime) and the wall-clock rate; 512-thread workgroups, one per CU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "probe_bodies.inc"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define ACLOB                                                                                                          \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", \
        "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", \
        "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", \
        "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", \
        "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", \
        "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", \
        "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111",   \
        "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", \
        "a126", "a127"

#define PROBE_KERNEL(LEVEL)                                                                                              \
    __global__ __launch_bounds__(512, 2) void probe##LEVEL(const u32x4* in, unsigned long long* out, int iters) {        \
        extern __shared__ __attribute__((aligned(16))) char smem[];                                                      \
        u32x4* l = reinterpret_cast<u32x4*>(smem);                                                                       \
        for (int i = threadIdx.x; i < 4096; i += 512) l[i] = in[i];                                                      \
        __syncthreads();                                                                                                 \
        u32x4 x0 = in[threadIdx.x], x1 = in[threadIdx.x + 512], t0 = in[threadIdx.x + 1024], t1 = in[threadIdx.x + 1536]; \
        u32x4 t2 = in[threadIdx.x + 2048], t3 = in[threadIdx.x + 2560], q0 = in[threadIdx.x + 3072], q1 = in[threadIdx.x + 3584]; \
        unsigned ldr = (threadIdx.x & 63) * 16, ldw = (threadIdx.x >> 6) * 2048 + (threadIdx.x & 63) * 16 + 0x8000;      \
        unsigned voff = threadIdx.x * 16;                                                                                \
        u32x4 srd;                                                                                                       \
        const unsigned long long b = (unsigned long long)in;                                                             \
        srd[0] = (unsigned)b; srd[1] = (unsigned)(b >> 32) & 0xffff; srd[2] = 65536; srd[3] = 0x00020000;                \
        srd[0] = __builtin_amdgcn_readfirstlane(srd[0]); srd[1] = __builtin_amdgcn_readfirstlane(srd[1]);                \
        unsigned long long ta, tb;                                                                                       \
        asm volatile("s_memtime %[ta]\n\ts_waitcnt lgkmcnt(0)\n\t"                                                       \
                     ".Lloop%=:\n\t" PROBE_BODY_##LEVEL                                                                  \
                     "s_sub_u32 %[n], %[n], 1\n\t"                                                                       \
                     "s_cmp_lg_u32 %[n], 0\n\t"                                                                          \
                     "s_cbranch_scc1 .Lloop%=\n\t"                                                                       \
                     "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_memtime %[tb]\n\ts_waitcnt lgkmcnt(0)\n\t"          \
                     : [ta] "=&s"(ta), [tb] "=&s"(tb), [n] "+s"(iters), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2),     \
                       [t3] "+v"(t3), [q0] "+v"(q0), [q1] "+v"(q1), [ldr] "+v"(ldr), [ldw] "+v"(ldw)                     \
                     : [x0] "v"(x0), [x1] "v"(x1), [voff] "v"(voff), [srd] "s"(srd)                                      \
                     : "memory", "scc", ACLOB);                                                                          \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + threadIdx.x / 64] = tb - ta;                                   \
    }

PROBE_KERNEL(0)
PROBE_KERNEL(1)
PROBE_KERNEL(2)
PROBE_KERNEL(3)
PROBE_KERNEL(4)
PROBE_KERNEL(5)
PROBE_KERNEL(6)

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const int iters = 4000;  // chunks per wave
    std::vector<unsigned> h(4096 * 4);
    srand(7);
    for (auto& v : h) {
        auto bf = []() { return (unsigned)(((rand() & 1) << 15) | ((125 + (rand() & 3)) << 7) | (rand() & 127)); };
        v = bf() | (bf() << 16);
    }
    u32x4* din;
    unsigned long long* dout;
    CK(hipMalloc(&din, h.size() * 4));
    CK(hipMalloc(&dout, 256 * 8 * 8));
    CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    typedef void (*kern_t)(const u32x4*, unsigned long long*, int);
    kern_t ks[] = {probe0, probe1, probe2, probe3, probe4, probe5, probe6};
    const char* names[] = {"0 bare MFMA", "1 +rotating operands", "2 +s_waitcnt", "3 +ds_read_b128", "4 +s_barrier/chunk",
                           "5 +Q ds_write", "6 +Q buffer_load"};
    for (int rep = 0; rep < 2; ++rep)
        for (int k = 0; k < 7; ++k) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ks[k]), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
            float ms;
            CK(hipEventRecord(e0));
            ks[k]<<<256, 512, 65536>>>(din, dout, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> t(256 * 8);
            CK(hipMemcpy(t.data(), dout, t.size() * 8, hipMemcpyDeviceToHost));
            double sum = 0;
            for (auto v : t) sum += (double)v;
            const double cyc = sum / t.size();
            const double mfma_per_simd = (double)iters * 64 * 2;
            printf("%-24s: %.2f cycles per MFMA per SIMD, wall %.3f ms, %.2f PFLOP/s, clock %.2f GHz\n", names[k],
                   cyc / mfma_per_simd, ms, 1024.0 * mfma_per_simd * 16384 / (ms * 1e-3) / 1e15, cyc / (ms * 1e-3) / 1e9);
        }
    return 0;
}
