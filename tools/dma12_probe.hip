// gfx950 probe: global_load_lds_dwordx3 (12 bytes per lane) next to global_load_lds_dwordx4 in a counted-vmcnt ring, as the
// bf6w body stream would use it (a wave's 5,632-byte share of a 22 KiB chunk = 4 x 1 KiB + 2 x 768 B).  Checks (1) where the
// 12-byte pieces land (M0 + offset + lane * 12?), (2) whether `s_waitcnt vmcnt(N)` + barrier certifies them like the 16-byte
// ones when further pieces are in flight.  Every byte of every round is compared.
//   hipcc --offload-arch=gfx950 -O2 tools/dma12_probe.hip -o tools/dma12_probe && tools/dma12_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHUNK 22528
#define SHARE 5632
#define SLOT 28672
#define NSLOT 4

// mode 0: pieces 4, 5 as dwordx3; mode 1: the same bytes as 6 x dword (256 B each) for comparison
// one piece of 12 bytes per lane into zeroed LDS: where does lane l's data go?
__global__ void k_where(const char* src, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 2048 / 4; i += 64) reinterpret_cast<uint32_t*>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned off12 = threadIdx.x * 12;
    asm volatile("s_mov_b32 m0, 0\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %0, %1 offset:0\n\ts_waitcnt vmcnt(0)" :: "v"(off12), "s"(src) : "memory", "m0");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048 / 4; i += 64) out[i] = reinterpret_cast<uint32_t*>(lds)[i];
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_ring(const char* src, int n_chunks, unsigned* bad, unsigned* first_bad) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned off16 = wave * SHARE + lane * 16, off12 = wave * SHARE + 4096 + lane * 12, off4 = wave * SHARE + 4096 + lane * 4;
    auto issue = [&](int c) {
        const char* g = src + (size_t)(blockIdx.x * 7 + c) % 64 * CHUNK;     // 64 different chunks, per-block phase
        const unsigned m0 = (c % NSLOT) * SLOT + wave * SHARE;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t" :: "s"(m0), "v"(off16), "s"(g) : "memory", "m0");
        if (MODE == 0)
            asm volatile("global_load_lds_dwordx3 %0, %1 offset:0\n\tglobal_load_lds_dwordx3 %0, %1 offset:768\n\t" :: "v"(off12), "s"(g) : "memory");
        else
            asm volatile("global_load_lds_dword %0, %1 offset:0\n\tglobal_load_lds_dword %0, %1 offset:256\n\t"
                         "global_load_lds_dword %0, %1 offset:512\n\tglobal_load_lds_dword %0, %1 offset:768\n\t"
                         "global_load_lds_dword %0, %1 offset:1024\n\tglobal_load_lds_dword %0, %1 offset:1280\n\t" :: "v"(off4), "s"(g) : "memory");
    };
    unsigned nbad = 0, fb = 0xffffffffu;
    issue(0); issue(1); issue(2);
    for (int c = 0; c < n_chunks; ++c) {
        // certify chunk c: all but the two youngest chunks' pieces of this wave are done
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* g = src + (size_t)(blockIdx.x * 7 + c) % 64 * CHUNK;
        const uint32_t* want = reinterpret_cast<const uint32_t*>(g);
        const uint32_t* got = reinterpret_cast<const uint32_t*>(lds + (c % NSLOT) * SLOT);
        for (int i = threadIdx.x; i < CHUNK / 4; i += 256)
            if (got[i] != want[i]) { ++nbad; if (fb == 0xffffffffu) fb = (unsigned)c << 16 | (unsigned)(i & 0xffff); }
        __builtin_amdgcn_s_barrier();                    // everybody has read slot c % 4 ...
        if (c + 3 < n_chunks) issue(c + 3);              // ... before chunk c + 3 (another slot) goes out; slot of c is reused by c + 4
        else asm volatile("s_nop 0");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (nbad) { atomicAdd(bad, nbad); atomicMin(first_bad, fb); }
}

int main() {
    const size_t n = 64 * (size_t)CHUNK;
    char* h = (char*)malloc(n);
    srand(5);
    for (size_t i = 0; i < n; ++i) h[i] = (char)rand();
    char* d; unsigned *bad, *fb;
    hipMalloc((void**)&d, n); hipMalloc((void**)&bad, 4); hipMalloc((void**)&fb, 4);
    hipMemcpy(d, h, n, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring<0>), hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring<1>), hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT);
    {
        uint32_t* dout; hipMalloc((void**)&dout, 2048);
        k_where<<<1, 64, 2048>>>(d, dout);
        uint32_t got[512];
        hipMemcpy(got, dout, 2048, hipMemcpyDeviceToHost);
        const uint32_t* w = reinterpret_cast<const uint32_t*>(h);
        printf("global_load_lds_dwordx3, M0 = 0, lane l reads source bytes [12 l, 12 l + 12): LDS dword i holds source dword:");
        for (int i = 0; i < 40; ++i) {
            int k = -1;
            for (int j = 0; j < 192; ++j) if (w[j] == got[i]) k = j;
            if (got[i] == 0xdeadbeefu) printf(" -"); else printf(" %d", k);
        }
        int n16 = 0, n12 = 0;
        for (int l = 0; l < 64; ++l) for (int k = 0; k < 3; ++k) { n16 += got[4 * l + k] == w[3 * l + k]; n12 += got[3 * l + k] == w[3 * l + k]; }
        printf("\n  matches if lane l lands at 16 l: %d of 192; at 12 l: %d of 192\n", n16, n12);
    }
    for (int mode = 0; mode < 2; ++mode) {
        unsigned z = 0, f = 0xffffffffu;
        hipMemcpy(bad, &z, 4, hipMemcpyHostToDevice); hipMemcpy(fb, &f, 4, hipMemcpyHostToDevice);
        const int chunks = 2000;
        if (mode == 0) k_ring<0><<<256, 256, NSLOT * SLOT>>>(d, chunks, bad, fb); else k_ring<1><<<256, 256, NSLOT * SLOT>>>(d, chunks, bad, fb);
        hipDeviceSynchronize();
        hipMemcpy(&z, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&f, fb, 4, hipMemcpyDeviceToHost);
        printf("%s: %u wrong dwords of %.3g (256 workgroups x %d chunks of 22 KiB, 2 chunks in flight)%s\n",
               mode == 0 ? "4 x dwordx4 + 2 x dwordx3 per wave, vmcnt(12)" : "4 x dwordx4 + 6 x dword per wave,   vmcnt(20)", z,
               256.0 * chunks * CHUNK / 4, z ? "" : "  -> lands at M0 + offset + lane x size, certified by the counted wait + barrier");
        if (z) printf("   first wrong: chunk %u dword %u (byte %u of the chunk = byte %u of wave %u's share)\n", f >> 16, f & 0xffff,
                      (f & 0xffff) * 4, ((f & 0xffff) * 4) % SHARE, ((f & 0xffff) * 4) / SHARE);
    }
    return 0;
}
