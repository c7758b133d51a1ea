// host_math.h -- host-side number theory used while building a context (table generation,
// plaintext lifting, CRT constants).  Product code; independent of oracle/.
#pragma once
#include <stdint.h>
#include <vector>

namespace hostmath {

typedef unsigned __int128 u128;
typedef unsigned long long u64;

inline u64 mulmod(u64 a, u64 b, u64 m) { return (u64)(((u128)a * b) % m); }
inline u64 addmod(u64 a, u64 b, u64 m) { u64 s = a + b; return (s >= m || s < a) ? s - m : s; }
inline u64 submod(u64 a, u64 b, u64 m) { return a >= b ? a - b : a + m - b; }
inline u64 powmod(u64 a, u64 e, u64 m) {
    u64 r = 1 % m;
    a %= m;
    for (; e; e >>= 1) {
        if (e & 1) r = mulmod(r, a, m);
        a = mulmod(a, a, m);
    }
    return r;
}
inline u64 invmod(u64 a, u64 prime) { return powmod(a, prime - 2, prime); }

inline bool is_prime(u64 n) {
    if (n < 2) return false;
    const u64 bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (u64 p : bases) {
        if (n == p) return true;
        if (n % p == 0) return false;
    }
    u64 d = n - 1;
    int s = 0;
    while (!(d & 1)) { d >>= 1; ++s; }
    for (u64 a : bases) {
        u64 x = powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool witness = true;
        for (int r = 1; r < s && witness; ++r) {
            x = mulmod(x, x, n);
            if (x == n - 1) witness = false;
        }
        if (witness) return false;
    }
    return true;
}

inline int bit_length(u64 v) { return v ? 64 - __builtin_clzll(v) : 0; }
inline uint32_t bit_reverse(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i, x >>= 1) r = (r << 1) | (x & 1);
    return r;
}
// a primitive 2n-th root of unity modulo prime q (q = 1 mod 2n); 0 if none found
inline u64 primitive_2n_root(u64 q, u64 n) {
    for (u64 g = 2; g < 4096; ++g) {
        u64 c = powmod(g, (q - 1) / (2 * n), q);
        if (powmod(c, n, q) == q - 1) return c;
    }
    return 0;
}
inline u64 shoup(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }

// residue of a product of word-sized factors, skipping index `skip` (use skip = -1 for all)
inline u64 prod_mod(const u64 *f, int count, int skip, u64 m) {
    u64 r = 1 % m;
    for (int i = 0; i < count; ++i)
        if (i != skip) r = mulmod(r, f[i] % m, m);
    return r;
}

// little-endian multi-word unsigned integers, just enough for CRT composition and t*x/q rounding
struct BigUInt {
    std::vector<u64> w;
    explicit BigUInt(u64 v = 0, size_t words = 12) : w(words, 0) { w[0] = v; }
    void mul_small(u64 s) {
        u128 c = 0;
        for (auto &x : w) { c += (u128)x * s; x = (u64)c; c >>= 64; }
    }
    void add(const BigUInt &o) {
        u128 c = 0;
        for (size_t i = 0; i < w.size(); ++i) { c += (u128)w[i] + o.w[i]; w[i] = (u64)c; c >>= 64; }
    }
    void sub(const BigUInt &o) {   // requires *this >= o
        u64 borrow = 0;
        for (size_t i = 0; i < w.size(); ++i) {
            u128 d = (u128)w[i] - o.w[i] - borrow;
            w[i] = (u64)d;
            borrow = (u64)(d >> 64) & 1;
        }
    }
    int cmp(const BigUInt &o) const {
        for (size_t i = w.size(); i-- > 0;)
            if (w[i] != o.w[i]) return w[i] > o.w[i] ? 1 : -1;
        return 0;
    }
    u64 mod_small(u64 m) const {
        u128 r = 0;
        for (size_t i = w.size(); i-- > 0;) r = ((r << 64) | w[i]) % m;
        return (u64)r;
    }
    int bits() const {
        for (size_t i = w.size(); i-- > 0;)
            if (w[i]) return (int)(64 * i) + bit_length(w[i]);
        return 0;
    }
    void shr1() {
        for (size_t i = 0; i < w.size(); ++i) w[i] = (w[i] >> 1) | (i + 1 < w.size() ? w[i + 1] << 63 : 0);
    }
};

}  // namespace hostmath
