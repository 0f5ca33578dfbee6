// ASan + UBSan driver of the library's HOST-ONLY code (SURVEY 5 planned -fsanitize=address for the host
// build): the packers that turn state_dict tensors into the MFMA weight streams (pack_image_host / pack_head_v1 /
// pack_body_v3 in csrc/r2l_capi.hip, pack_chain in csrc/nerf_capi.hip: ~600 lines of index arithmetic writing into byte
// vectors) and the numpy-shuffle restatement (csrc/np_shuffle.hip), linked from the same sources compiled host-only
// (`make -C efficient-nerf_amd/csrc asan`; no device code, no GPU call is made).  tests/test_packing_cpu.py feeds it seeded
// weights and compares the FNV-1a digests it prints with those of the product library's packers on the same inputs, and
// requires a clean sanitizer log.
//   pack_asan <weights.bin> : file = int32 n_block, then the R2L state_dict tensors (f32, state_dict order), then the 24
//                             teacher tensors
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../include/r2l_hip.h"

static uint64_t fnv(const std::vector<char>& b) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : b) h = (h ^ c) * 1099511628211ull;
    return h;
}

static const size_t kTeacher[24] = {256 * 63, 256, 256 * 256, 256, 256 * 256, 256, 256 * 256, 256, 256 * 256, 256, 256 * 319, 256,
                                    256 * 256, 256, 256 * 256, 256, 128 * 283, 128, 256 * 256, 256, 256, 1, 3 * 128, 3};

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int n_block = 0;
    if (fread(&n_block, 4, 1, f) != 1 || n_block < 0 || n_block > 64) return 2;
    std::vector<std::vector<float>> w;
    auto rd = [&](size_t n) {
        w.emplace_back(n);
        if (fread(w.back().data(), 4, n, f) != n) exit(2);
    };
    rd(256 * 1008);
    rd(256);
    for (int b = 0; b < n_block; ++b)
        for (int l = 0; l < 2; ++l) {
            rd(256 * 256);
            rd(256);
        }
    rd(3 * 256);
    rd(3);
    const int n_r2l = (int)w.size();
    for (size_t n : kTeacher) rd(n);
    fclose(f);
    std::vector<const float*> p;
    for (auto& v : w) p.push_back(v.data());
    for (int mode = 0; mode <= 6; ++mode) {       // R2L_PREC_FP16X3, FP16X1, FP16_FP8, FP16_E4M3, FP16X3_ASM, FP16_SPLIT, FP16_SPLIT8: chunk / head streams
        long long n = r2l_debug_pack_host(p.data(), n_r2l, n_block, mode, nullptr, 0);
        if (n <= 0) {
            printf("pack_host mode %d: %s\n", mode, r2l_last_error());
            return 1;
        }
        std::vector<char> out((size_t)n);
        if (r2l_debug_pack_host(p.data(), n_r2l, n_block, mode, out.data(), n) != n) return 1;
        printf("pack_host %d %lld %016llx\n", mode, n, (unsigned long long)fnv(out));
    }
    for (int fmt : {0, 1, 3}) {                    // body streams: bf6 terms, e4m3 terms, three fp16 passes
        if (r2l_debug_pack_body_format(fmt)) return 1;
        long long offs[2] = {0, 0};
        long long n = r2l_debug_pack_body_host(p.data(), n_r2l, n_block, nullptr, 0, offs);
        if (n <= 0) {
            printf("pack_body fmt %d: %s\n", fmt, r2l_last_error());
            return 1;
        }
        std::vector<char> out((size_t)n);
        if (r2l_debug_pack_body_host(p.data(), n_r2l, n_block, out.data(), n, offs) != n) return 1;
        printf("pack_body %d %lld %016llx %lld %lld\n", fmt, n, (unsigned long long)fnv(out), offs[0], offs[1]);
    }
    r2l_debug_pack_body_format(0);
    for (int fmt : {0, 1, 2, 3, 4, 5, 6}) {            // the teacher's chain streams: bf6 terms, one fp16 pass, three fp16 passes (hi | lo pieces), mix, three passes without the view branch
        long long off = 0;
        long long n = nerf_debug_pack_chain_host(p.data() + n_r2l, 24, fmt, nullptr, 0, &off);
        if (n <= 0) {
            printf("pack_chain fmt %d: %s\n", fmt, r2l_last_error());
            return 1;
        }
        std::vector<char> out((size_t)n);
        if (nerf_debug_pack_chain_host(p.data() + n_r2l, 24, fmt, out.data(), n, &off) != n) return 1;
        printf("pack_chain %d %lld %016llx %lld\n", fmt, n, (unsigned long long)fnv(out), off);
    }
    {
        std::vector<unsigned> key(624);
        for (int i = 0; i < 624; ++i) key[i] = 2654435761u * (unsigned)(i + 1);
        int pos = 300;
        for (long long n : {0ll, 1ll, 2ll, 63ll, 64ll, 65ll, 100000ll}) {
            std::vector<int> out((size_t)n + 1, -7);
            if (r2l_np_legacy_permutation(key.data(), &pos, n, out.data())) return 1;
            if (out[(size_t)n] != -7) return 1;    // nothing behind the n-th element is written
            std::vector<char> bytes((const char*)out.data(), (const char*)(out.data() + n));
            printf("perm %lld %016llx %d\n", n, (unsigned long long)fnv(bytes), pos);
        }
    }
    return 0;
}
