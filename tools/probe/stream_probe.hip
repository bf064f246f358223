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

#define ACLOB2 "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

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

// MT = 4: one wave per SIMD, 64 rows x 256 queries per wave in a[0:255]
__global__ __launch_bounds__(256, 1) void probe_mt4(const u32x4* in, unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* l = reinterpret_cast<u32x4*>(smem);
    for (int i = threadIdx.x; i < 4096; i += 256) l[i] = in[i];
    __syncthreads();
    u32x4 x0 = in[threadIdx.x], x1 = in[threadIdx.x + 512], t0 = in[threadIdx.x + 1024], t1 = in[threadIdx.x + 1536];
    u32x4 t2 = in[threadIdx.x + 2048], t3 = in[threadIdx.x + 2560], q0 = in[threadIdx.x + 3072], q1 = in[threadIdx.x + 3584];
    unsigned ldr = (threadIdx.x & 63) * 16, ldw = (threadIdx.x >> 6) * 2048 + (threadIdx.x & 63) * 16 + 0x8000;
    unsigned voff = threadIdx.x * 16;
    u32x4 srd;
    const unsigned long long b = (unsigned long long)in;
    srd[0] = (unsigned)b; srd[1] = (unsigned)(b >> 32) & 0xffff; srd[2] = 65536; srd[3] = 0x00020000;
    srd[0] = __builtin_amdgcn_readfirstlane(srd[0]); srd[1] = __builtin_amdgcn_readfirstlane(srd[1]);
    unsigned long long ta, tb;
    asm volatile("s_memtime %[ta]\n\ts_waitcnt lgkmcnt(0)\n\t"
                 ".Lloop%=:\n\t" PROBE_BODY_MT4
                 "s_sub_u32 %[n], %[n], 1\n\t"
                 "s_cmp_lg_u32 %[n], 0\n\t"
                 "s_cbranch_scc1 .Lloop%=\n\t"
                 "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_memtime %[tb]\n\ts_waitcnt lgkmcnt(0)\n\t"
                 : [ta] "=&s"(ta), [tb] "=&s"(tb), [n] "+s"(iters), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2),
                   [t3] "+v"(t3), [q0] "+v"(q0), [q1] "+v"(q1), [ldr] "+v"(ldr), [ldw] "+v"(ldw)
                 : [x0] "v"(x0), [x1] "v"(x1), [voff] "v"(voff), [srd] "s"(srd)
                 : "memory", "scc", ACLOB, ACLOB2);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = tb - ta;
}

// level 6 with LDS-DMA staging
__global__ __launch_bounds__(512, 2) void probe_dma(const u32x4* in, unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* l = reinterpret_cast<u32x4*>(smem);
    for (int i = threadIdx.x; i < 4096; i += 512) l[i] = in[i];
    __syncthreads();
    u32x4 x0 = in[threadIdx.x], x1 = in[threadIdx.x + 512], t0 = in[threadIdx.x + 1024], t1 = in[threadIdx.x + 1536];
    u32x4 t2 = in[threadIdx.x + 2048], t3 = in[threadIdx.x + 2560];
    unsigned ldr = (threadIdx.x & 63) * 16;
    unsigned sldw = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * 2048 + 0x8000);
    unsigned voff = threadIdx.x * 16;
    u32x4 srd;
    const unsigned long long b = (unsigned long long)in;
    srd[0] = (unsigned)b; srd[1] = (unsigned)(b >> 32) & 0xffff; srd[2] = 65536; srd[3] = 0x00020000;
    srd[0] = __builtin_amdgcn_readfirstlane(srd[0]); srd[1] = __builtin_amdgcn_readfirstlane(srd[1]);
    unsigned long long ta, tb;
    asm volatile("s_memtime %[ta]\n\ts_waitcnt lgkmcnt(0)\n\t"
                 ".Lloop%=:\n\t" PROBE_BODY_DMA
                 "s_sub_u32 %[n], %[n], 1\n\t"
                 "s_cmp_lg_u32 %[n], 0\n\t"
                 "s_cbranch_scc1 .Lloop%=\n\t"
                 "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_memtime %[tb]\n\ts_waitcnt lgkmcnt(0)\n\t"
                 : [ta] "=&s"(ta), [tb] "=&s"(tb), [n] "+s"(iters), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2),
                   [t3] "+v"(t3), [ldr] "+v"(ldr), [sldw] "+s"(sldw)
                 : [x0] "v"(x0), [x1] "v"(x1), [voff] "v"(voff), [srd] "s"(srd)
                 : "memory", "scc", "m0", ACLOB);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + threadIdx.x / 64] = tb - ta;
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
    kern_t ks[] = {probe0, probe1, probe2, probe3, probe4, probe5, probe6, probe_dma};
    const char* names[] = {"0 bare MFMA", "1 +rotating operands", "2 +s_waitcnt", "3 +ds_read_b128", "4 +s_barrier/chunk",
                           "5 +Q ds_write", "6 +Q buffer_load", "6d Q staged by LDS-DMA"};
    for (int rep = 0; rep < 2; ++rep)
        for (int k = 0; k < 8; ++k) {
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
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_mt4), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        float ms;
        CK(hipEventRecord(e0));
        probe_mt4<<<256, 256, 65536>>>(din, dout, iters / 2);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> t(256 * 4);
        CK(hipMemcpy(t.data(), dout, t.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (auto v : t) sum += (double)v;
        const double cyc = sum / t.size();
        const double mfma_per_simd = (double)(iters / 2) * 128;
        printf("%-24s: %.2f cycles per MFMA per SIMD, wall %.3f ms, %.2f PFLOP/s, clock %.2f GHz\n",
               "6' one wave/SIMD, MT=4", cyc / mfma_per_simd, ms, 1024.0 * mfma_per_simd * 16384 / (ms * 1e-3) / 1e15,
               cyc / (ms * 1e-3) / 1e9);
    }
    return 0;
}
