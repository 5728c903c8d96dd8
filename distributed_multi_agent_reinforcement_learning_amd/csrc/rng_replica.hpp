// rng_replica.hpp -- bit-exact host replicas of the generator streams the reference environments draw from:
// CPython's random.Random (MT19937 + init_by_array seeding + _randbelow_with_getrandbits) and numpy's legacy
// RandomState (MT19937 + init_genrand seeding, random_sample, polar Box-Muller legacy_gauss with its cached value).
#pragma once
#include <math.h>
#include <stdint.h>

namespace rngrep {

struct MT19937 {
    uint32_t mt[624];
    int idx;
    void init_genrand(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    void init_by_array(const uint32_t *key, int len) {
        init_genrand(19650218u);
        int i = 1, j = 0;
        int k = 624 > len ? 624 : len;
        for (; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            i++; j++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
            if (j >= len) j = 0;
        }
        for (k = 623; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            i++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
        idx = 624;
    }
    uint32_t next() {
        if (idx >= 624) {
            for (int k = 0; k < 624; k++) {
                uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};

// CPython random.Random seeded with a non-negative int (Lib/random.py seed -> _randommodule.c init_by_array)
struct PyRandom {
    MT19937 g;
    void seed(uint64_t a) {
        uint32_t key[2] = {(uint32_t)a, (uint32_t)(a >> 32)};
        g.init_by_array(key, key[1] ? 2 : 1);
    }
    uint32_t getrandbits(int k) { return g.next() >> (32 - k); }  // 1 <= k <= 32
    uint32_t randbelow(uint32_t n) {                              // _randbelow_with_getrandbits
        int k = 0;
        for (uint32_t v = n; v; v >>= 1) k++;
        uint32_t r = getrandbits(k);
        while (r >= n) r = getrandbits(k);
        return r;
    }
    int randint(int a, int b) { return a + (int)randbelow((uint32_t)(b - a + 1)); }
};

// numpy.random.RandomState (legacy) seeded with a 32-bit int: rand(), normal()
struct NpRandom {
    MT19937 g;
    bool has_gauss = false;
    double gauss = 0.0;
    void seed(uint32_t s) { g.init_genrand(s); has_gauss = false; gauss = 0.0; }
    double random_sample() {
        uint32_t a = g.next() >> 5, b = g.next() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    double legacy_gauss() {
        if (has_gauss) { has_gauss = false; double t = gauss; gauss = 0.0; return t; }
        double f, x1, x2, r2;
        do {
            x1 = 2.0 * random_sample() - 1.0;
            x2 = 2.0 * random_sample() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        f = sqrt(-2.0 * log(r2) / r2);
        gauss = f * x1;
        has_gauss = true;
        return f * x2;
    }
    double normal(double loc, double scale) { return loc + scale * legacy_gauss(); }
};

}  // namespace rngrep
