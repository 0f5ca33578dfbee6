// Does rewriting M0 (or the address VGPR / SGPR pair) in the instruction slot directly behind a global_load_lds corrupt
// that LDS-DMA on gfx950?  (Round-1 commit 4d9e2ee issued its LDS-DMA from inline asm in exactly that form and "faulted
// intermittently"; the fault was never root-caused.)  Every variant below copies 4 KiB per wave global -> LDS in four
// 1 KiB pieces and is checked byte for byte; every address any variant could form -- old or new value of M0 / of the
// address register -- stays inside the buffers, so a hazard shows up as wrong LDS contents, never as a fault.
//   A  reference form: M0 and the address VGPR advance per piece, 16 wait states (2 x s_nop 7) behind each DMA first
//   B  M0 rewritten (s_add_u32 m0, m0, 0x400) in the slot right behind each DMA          (the 4d9e2ee form)
//   C  M0 AND the address VGPR rewritten in the two slots right behind each DMA           (the 4d9e2ee form)
//   D  saddr form (SGPR pair + 32-bit VGPR offset), M0 restored right behind the last DMA (the 4d9e2ee form)
// hipcc --offload-arch=gfx950 -O2 tools/m0_hazard_test.hip -o tools/m0_hazard_test && tools/m0_hazard_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

extern __shared__ __attribute__((aligned(16))) char smem[];

template <int V>
__global__ __launch_bounds__(256) void k(const char* src, uint32_t* out, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t lds0 = wave * 4096;
    uint32_t bad = 0;
    for (int it = 0; it < iters; ++it) {
        // a different 4 KiB window every iteration and workgroup (64 KiB source buffer + 8 KiB slack)
        const uint32_t goff = (((blockIdx.x * 131u + it * 17u) & 15u) * 4096u) + wave * 4096u % 16384u;
        const char* base = src + goff;
        uint32_t keep, tv;
        const uint32_t voff = lane * 16;
        if (V == 0) {
            asm volatile(
                "s_nop 4\n\ts_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %[voff], %[src]\n\ts_nop 7\n\ts_nop 7\n\t"
                "s_add_u32 m0, m0, 0x400\n\tv_add_u32 %[tv], 0x400, %[voff]\n\ts_nop 1\n\t"
                "global_load_lds_dwordx4 %[tv], %[src]\n\ts_nop 7\n\ts_nop 7\n\t"
                "s_add_u32 m0, m0, 0x400\n\tv_add_u32 %[tv], 0x800, %[voff]\n\ts_nop 1\n\t"
                "global_load_lds_dwordx4 %[tv], %[src]\n\ts_nop 7\n\ts_nop 7\n\t"
                "s_add_u32 m0, m0, 0x400\n\tv_add_u32 %[tv], 0xc00, %[voff]\n\ts_nop 1\n\t"
                "global_load_lds_dwordx4 %[tv], %[src]\n\ts_nop 7\n\ts_nop 7\n\t"
                "s_mov_b32 m0, %[keep]"
                : [keep] "=&s"(keep), [tv] "=&v"(tv) : [dst] "s"(lds0), [voff] "v"(voff), [src] "s"(base) : "memory", "scc");
        } else if (V == 1) {
            uint32_t t1 = voff + 0x400, t2 = voff + 0x800, t3 = voff + 0xc00;
            asm volatile(
                "s_nop 4\n\ts_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %[voff], %[src]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %[t1], %[src]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %[t2], %[src]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %[t3], %[src]\n\ts_mov_b32 m0, %[keep]"
                : [keep] "=&s"(keep) : [dst] "s"(lds0), [voff] "v"(voff), [t1] "v"(t1), [t2] "v"(t2), [t3] "v"(t3), [src] "s"(base)
                : "memory", "scc");
        } else if (V == 2) {
            asm volatile(
                "s_nop 4\n\ts_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\t"
                "v_add_u32 %[tv], 0x400, %[voff]\n\t"
                "global_load_lds_dwordx4 %[voff], %[src]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %[tv], %[src]\n\ts_add_u32 m0, m0, 0x400\n\tv_add_u32 %[tv], 0x800, %[voff]\n\t"
                "global_load_lds_dwordx4 %[tv], %[src]\n\ts_add_u32 m0, m0, 0x400\n\tv_add_u32 %[tv], 0xc00, %[voff]\n\t"
                "global_load_lds_dwordx4 %[tv], %[src]\n\ts_mov_b32 m0, %[keep]"
                : [keep] "=&s"(keep), [tv] "=&v"(tv) : [dst] "s"(lds0), [voff] "v"(voff), [src] "s"(base) : "memory", "scc");
        } else {
            asm volatile(
                "s_nop 4\n\ts_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %[voff], %[src]\n\t"
                "global_load_lds_dwordx4 %[voff], %[src] offset:1024\n\t"
                "global_load_lds_dwordx4 %[voff], %[src] offset:2048\n\t"
                "global_load_lds_dwordx4 %[voff], %[src] offset:3072\n\ts_mov_b32 m0, %[keep]"
                : [keep] "=&s"(keep) : [dst] "s"(lds0), [voff] "v"(voff), [src] "s"(base) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int p = 0; p < 4; ++p) {
            const uint4 got = *reinterpret_cast<const uint4*>(smem + lds0 + p * 1024 + lane * 16);
            const uint4 want = *reinterpret_cast<const uint4*>(base + p * 1024 + lane * 16);
            bad += (got.x != want.x) + (got.y != want.y) + (got.z != want.z) + (got.w != want.w);
        }
        __syncthreads();
        *reinterpret_cast<uint4*>(smem + lds0 + (it & 3) * 1024 + lane * 16) = uint4{0xdeadbeefu, 0u, 0u, 0u};  // poison one piece
        __syncthreads();
    }
    if (bad) atomicAdd(out, bad);
}

int main() {
    const size_t N = 65536 + 32768;
    std::vector<uint32_t> h(N / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) ^ 0x5bd1e995u;
    char* d; uint32_t* dbad;
    hipMalloc((void**)&d, N); hipMalloc((void**)&dbad, 4);
    hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
    const char* names[4] = {"A padded reference", "B M0 rewritten right behind each DMA", "C M0 and address VGPR rewritten right behind each DMA",
                            "D saddr form, M0 restored right behind the last DMA"};
    int fails = 0;
    for (int v = 0; v < 4; ++v) {
        hipMemset(dbad, 0, 4);
        const int iters = 2000;
        if (v == 0) k<0><<<1024, 256, 16384>>>(d, dbad, iters);
        if (v == 1) k<1><<<1024, 256, 16384>>>(d, dbad, iters);
        if (v == 2) k<2><<<1024, 256, 16384>>>(d, dbad, iters);
        if (v == 3) k<3><<<1024, 256, 16384>>>(d, dbad, iters);
        hipError_t e = hipDeviceSynchronize();
        uint32_t bad = 0;
        hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
        printf("%-55s: %s, %u wrong dwords of %llu (1024 workgroups x %d iterations)\n", names[v], hipGetErrorString(e), bad,
               1024ull * 4 * 64 * 16 * iters, iters);
        fails += bad != 0 || e != hipSuccess;
    }
    return fails;
}
