// Host-only: the shuffle of `create_data rand` (utils/create_data.py:858-859 of the reference: two
// `np.random.permutation(n)` per save group, n = 16,000,000 at the reference's sizes) restated so that the ONE serial
// piece of config 5 -- every rank has to walk the same global numpy stream -- costs 0.1 s per permutation instead of
// numpy's 0.5-2.4 s.  The algorithm is numpy's legacy `RandomState.permutation` (numpy is a third-party dependency of the
// reference, requirements.txt:9 pins 1.22; the legacy stream is frozen by numpy's compatibility policy, NEP 19):
//   permutation(n)   = arange(n), then shuffle                                  (numpy/random/mtrand.pyx)
//   shuffle, 1-d     : for i = n-1 .. 1:  j = random_interval(i);  swap(x[i], x[j])          (_shuffle_raw)
//   random_interval  : mask = smallest 2^k - 1 >= max; draw 32-bit words & mask until <= max (max <= 0xffffffff)
//                                                                        (numpy/random/src/distributions: legacy interval)
//   32-bit words     : MT19937 with the standard tempering (mt19937_next32 / mt19937_gen)
// The state travels through RandomState.get_state() / set_state() (624 key words + position), so draws before and after --
// the pose angles and focal factors, `rs.rand()` -- stay numpy's own.  tests/test_create_data_cpu.py checks the result and the
// state left behind against numpy bit for bit.
#include <stdint.h>

#include "../../include/r2l_hip.h"
#include "r2l_host_util.h"

namespace {
constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;

struct MT {
    uint32_t* key;
    int pos;
    void gen() {
        int kk = 0;
        uint32_t y;
        for (; kk < MT_N - MT_M; ++kk) {
            y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + MT_M] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX_A);
        }
        for (; kk < MT_N - 1; ++kk) {
            y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX_A);
        }
        y = (key[MT_N - 1] & UPPER) | (key[0] & LOWER);
        key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX_A);
        pos = 0;
    }
    inline uint32_t next32() {
        if (pos == MT_N) gen();
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};
}  // namespace

int r2l_np_legacy_permutation(unsigned* mt_key624, int* mt_pos, long long n, int* out) {
    if (!mt_key624 || !mt_pos || !out) return r2l_set_error(R2L_EINVAL, "r2l_np_legacy_permutation: NULL argument");
    if (n < 0 || n > 0x7fffffffll) return r2l_set_error(R2L_EINVAL, "r2l_np_legacy_permutation: n = %lld is outside [0, 2^31)", n);
    if (*mt_pos < 0 || *mt_pos > MT_N) return r2l_set_error(R2L_EINVAL, "r2l_np_legacy_permutation: MT19937 position %d is outside [0, 624]", *mt_pos);
    MT mt{mt_key624, *mt_pos};
    for (long long i = 0; i < n; ++i) out[i] = (int)i;
    // j depends on the random stream only, never on the array's contents: the draws run AHEAD batches of swaps and the
    // cache lines of out[j] (random addresses in a 64 MB array) are prefetched while the previous batch is being swapped
    constexpr int AHEAD = 64;
    uint32_t jq[2][AHEAD];
    uint32_t mask = 0;
    long long i_draw = n - 1;
    auto draw_batch = [&](uint32_t* q) -> int {
        int k = 0;
        for (; k < AHEAD && i_draw >= 1; ++k, --i_draw) {
            // smallest bit mask >= i: recomputed only when i drops below a power of two
            if (mask == 0 || (uint32_t)i_draw <= (mask >> 1)) {
                mask = (uint32_t)i_draw;
                mask |= mask >> 1;
                mask |= mask >> 2;
                mask |= mask >> 4;
                mask |= mask >> 8;
                mask |= mask >> 16;
            }
            uint32_t j;
            do {
                j = mt.next32() & mask;
            } while (j > (uint32_t)i_draw);
            q[k] = j;
            __builtin_prefetch(&out[j], 1, 0);
        }
        return k;
    };
    int cur = 0;
    int have = draw_batch(jq[cur]);
    long long i = n - 1;
    while (have > 0) {
        const int next_have = draw_batch(jq[cur ^ 1]);
        const uint32_t* q = jq[cur];
        for (int k = 0; k < have; ++k, --i) {
            const uint32_t j = q[k];
            const int t = out[i];
            out[i] = out[j];
            out[j] = t;
        }
        cur ^= 1;
        have = next_have;
    }
    *mt_pos = mt.pos;
    return R2L_OK;
}
