// Probe: how fast can every CU stream the SAME weight image (24 MB, re-read per tile) into LDS?
//   mode 0: LDS-DMA (global_load_lds_dwordx4), 4-slot ring of 32 KiB chunks, counted vmcnt, one barrier per chunk
//   mode 1: global_load_dwordx4 -> registers (two chunks ahead) -> ds_write_b128, one barrier per chunk
//   mode 2: global_load_dwordx4 only (data consumed by a v_xor), no LDS
// 4 waves per CU (256 threads), no MFMA.  hipcc --offload-arch=gfx950 -O3 tools/stream_test.hip -o tools/stream_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) char smem[];
#define AS1(p) ((const __attribute__((address_space(1))) void*)(p))
#define AS3(p) ((__attribute__((address_space(3))) void*)(p))
#define CH 32768
#define NSLOT 4

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const char* __restrict__ w, int n_chunks, int passes, int* out) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int acc = 0;
    const int total = n_chunks * passes;
    if (MODE == 0) {
        int issue = 0;
        auto issue_chunk = [&](int c) {
            const char* src = w + (size_t)(c % n_chunks) * CH + wave * 8192 + lane * 16;
            const uint32_t dst = (uint32_t)(c % NSLOT) * CH + wave * 8192;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_global_load_lds(AS1(src + i * 4096), AS3(smem + dst + i * 4096), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(AS1(src + i * 4096), AS3(smem + dst + i * 4096), 16, 1024, 0);
                __builtin_amdgcn_global_load_lds(AS1(src + i * 4096), AS3(smem + dst + i * 4096), 16, 2048, 0);
                __builtin_amdgcn_global_load_lds(AS1(src + i * 4096), AS3(smem + dst + i * 4096), 16, 3072, 0);
            }
        };
        for (; issue < 3; ++issue) issue_chunk(issue);
        for (int c = 0; c < total; ++c) {
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // chunk c landed (c+1, c+2 may be in flight)
            __builtin_amdgcn_s_barrier();
            issue_chunk(issue++);                                // into the slot of chunk c-1
            acc ^= *reinterpret_cast<const int*>(smem + (c % NSLOT) * CH + threadIdx.x * 64);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        i32x4 st[2][8];
        auto load_chunk = [&](int c, i32x4 (&r)[8]) {
            const char* src = w + (size_t)(c % n_chunks) * CH + wave * 8192 + lane * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = *reinterpret_cast<const i32x4*>(src + i * 1024);
        };
        load_chunk(0, st[0]);
        load_chunk(1, st[1]);
        for (int c = 0; c < total; c += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (MODE == 1) {
                    const uint32_t dst = (uint32_t)((c + h) % NSLOT) * CH + wave * 8192 + lane * 16;
#pragma unroll
                    for (int i = 0; i < 8; ++i) *reinterpret_cast<i32x4*>(smem + dst + i * 1024) = st[h][i];
                    __builtin_amdgcn_s_barrier();
                    acc ^= *reinterpret_cast<const int*>(smem + ((c + h) % NSLOT) * CH + threadIdx.x * 64);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc ^= st[h][i][0];
                }
                load_chunk(c + h + 2, st[h]);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const int n_chunks = 744;  // 24.4 MB
    char* w; int* o;
    hipMalloc(&w, (size_t)(n_chunks + 4) * CH); hipMemset(w, 1, (size_t)(n_chunks + 4) * CH); hipMalloc(&o, 256 * 256 * 4);
    const int passes = 10;
    for (int mode = 0; mode < 3; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
#define L(M) hipFuncSetAttribute((const void*)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * CH); hipLaunchKernelGGL(k<M>, dim3(256), dim3(256), NSLOT * CH, 0, w, n_chunks, passes, o)
            if (mode == 0) { L(0); } else if (mode == 1) { L(1); } else { L(2); }
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double bytes = 256.0 * n_chunks * passes * CH;
        printf("mode %d: %.3f ms, %.2f TB/s aggregate, %.1f GB/s per CU\n", mode, ms, bytes / ms / 1e9, bytes / 256 / ms / 1e6);
    }
    return 0;
}
