#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// LDS-DMA semantics probe: wave 0 DMAs 1 KiB (64 lanes x 16 B) from global into LDS at byte offset `ldsoff`,
// then every thread dumps LDS to global.
__global__ void k(const unsigned* in, unsigned* out, unsigned ldsoff, unsigned instoff_sel) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0xdeadbeefu;
    __syncthreads();
    u32x4 srd;
    const unsigned long long b = (unsigned long long)in;
    srd[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffff);
    srd[2] = 65536; srd[3] = 0x00020000;
    unsigned voff = (threadIdx.x & 63) * 16;
    if (threadIdx.x < 64) {
        if (instoff_sel == 0)
            asm volatile("s_mov_b32 m0, %[m]\n\t"
                         "s_nop 0\n\t"
                         "buffer_load_dwordx4 %[voff], %[srd], 0 offen lds\n\t"
                         "s_waitcnt vmcnt(0)\n\t" :: [m] "s"(ldsoff), [voff] "v"(voff), [srd] "s"(srd) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %[m]\n\t"
                         "s_nop 0\n\t"
                         "buffer_load_dwordx4 %[voff], %[srd], 0 offen offset:2048 lds\n\t"
                         "s_waitcnt vmcnt(0)\n\t" :: [m] "s"(ldsoff), [voff] "v"(voff), [srd] "s"(srd) : "memory", "m0");
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) out[i] = lds[i];
}
int main() {
    unsigned *din, *dout; unsigned h[16384];
    for (int i = 0; i < 16384; ++i) h[i] = i;
    hipMalloc(&din, 65536); hipMalloc(&dout, 16384);
    hipMemcpy(din, h, 65536, hipMemcpyHostToDevice);
    for (int sel = 0; sel < 2; ++sel) {
        k<<<1, 256, 16384>>>(din, dout, 4096, sel);
        unsigned o[4096];
        hipMemcpy(o, dout, 16384, hipMemcpyDeviceToHost);
        int first = -1, last = -1;
        for (int i = 0; i < 4096; ++i) if (o[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; }
        printf("sel %d: LDS words changed [%d, %d]; first values %u %u %u %u %u; word at first+4: %u\n", sel, first, last, o[first], o[first+1], o[first+2], o[first+3], o[first+4], o[first+4]);
    }
    return 0;
}
