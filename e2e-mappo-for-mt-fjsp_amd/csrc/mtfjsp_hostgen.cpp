// mtfjsp_hostgen.cpp — host-side helper of the instance generator (instances.py): the two python loops of the reference's
// Instance_Dataset (instance/generate_allsize_mofjsp_dataset.py:204-216 "per task a uniform number k of machines made
// infeasible" and :241-272 "transport times") consume numpy's LEGACY RandomState stream one draw at a time — 2.4 minutes per
// rank at J20M20E4 x 16384.  Here the same draws are taken from the same MT19937 state at native speed, bit for bit:
//   randint(0, M)               = masked rejection sampling on 32-bit draws            (numpy legacy _rand_int64, use_masked)
//   choice(M, k, replace=False) = permutation(M)[:k] = Fisher-Yates from the top with random_interval(i)  (legacy _shuffle_raw)
//   uniform(lo, hi)             = lo + (hi - lo) * ((a >> 5) * 2^26 + (b >> 6)) / 2^53   (legacy random_uniform / next_double)
// The state (624 key words + position) comes from RandomState.get_state() and goes back with set_state(); rows outside
// [first, first + count) are drawn and dropped, so a rank generates only its shard's instances while the stream stays the
// reference's.  Pinned by tests/test_instances.py against the reference's own output (tests/golden/instances_generator.npz).
#include <cstdint>
#include <cstring>

namespace {
struct MT {
    uint32_t *key; int pos;
    void refill()
    {
        const int N = 624, Mm = 397;
        uint32_t y; int kk;
        for (kk = 0; kk < N - Mm; kk++) {
            y = (key[kk] & 0x80000000u) | (key[kk + 1] & 0x7fffffffu);
            key[kk] = key[kk + Mm] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
        }
        for (; kk < N - 1; kk++) {
            y = (key[kk] & 0x80000000u) | (key[kk + 1] & 0x7fffffffu);
            key[kk] = key[kk + (Mm - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
        }
        y = (key[N - 1] & 0x80000000u) | (key[0] & 0x7fffffffu);
        key[N - 1] = key[Mm - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
        pos = 0;
    }
    inline uint32_t next32()
    {
        if (pos == 624) refill();
        uint32_t y = key[pos++];
        y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
        return y;
    }
    inline double next_double()
    {
        const int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    inline uint32_t interval(uint32_t max)            // uniform on [0, max]
    {
        if (max == 0) return 0;
        uint32_t mask = max;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        while ((v = (next32() & mask)) > max) { }
        return v;
    }
};
}  // namespace

extern "C" {
// for every sample s and task row: k = randint(0, M); idx = choice(M, k, replace=False); t[s,row,idx] *= -1 — applied to the rows of
// samples [first, first + count) of `t` ([count, T, M], already holding the positive durations); other samples: drawn, dropped
int mtfjsp_hostgen_infeasible(uint32_t *key, int32_t *pos, int64_t S, int32_t T, int32_t M, int64_t first, int64_t count, double *t)
{
    if (!key || !pos || !t || M < 1 || M > 4096 || T < 1 || *pos < 0 || *pos > 624) return -1;
    MT mt{key, *pos};
    int64_t perm[4096];
    for (int64_t s = 0; s < S; s++) {
        const bool keep = s >= first && s < first + count;
        double *ts = keep ? t + (size_t)(s - first) * T * M : nullptr;
        for (int row = 0; row < T; row++) {
            const uint32_t k = M > 1 ? mt.interval((uint32_t)(M - 1)) : 0u;
            for (int i = 0; i < M; i++) perm[i] = i;
            for (int i = M - 1; i >= 1; i--) {
                const uint32_t j = mt.interval((uint32_t)i);
                const int64_t tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp;
            }
            if (keep) for (uint32_t i = 0; i < k; i++) ts[(size_t)row * M + perm[i]] *= -1.0;
        }
    }
    *pos = mt.pos;
    return 0;
}
// transport times: per sample an M x M matrix a[i][j] (i != j) = uniform(in_lo, in_hi) inside a shop, uniform(in_hi * d, out_hi * d)
// across shops at distance d, in (i, j) row-major draw order; tt = upper triangle mirrored (zero diagonal)
int mtfjsp_hostgen_transport(uint32_t *key, int32_t *pos, int64_t S, int32_t M, const int64_t *shop_of, double in_lo, double in_hi, double out_hi,
                             int64_t first, int64_t count, double *tt)
{
    if (!key || !pos || !tt || !shop_of || M < 1 || *pos < 0 || *pos > 624) return -1;
    MT mt{key, *pos};
    for (int64_t s = 0; s < S; s++) {
        const bool keep = s >= first && s < first + count;
        double *o = keep ? tt + (size_t)(s - first) * M * M : nullptr;
        if (keep) memset(o, 0, sizeof(double) * (size_t)M * M);
        for (int i = 0; i < M; i++)
            for (int j = 0; j < M; j++) {
                if (i == j) continue;
                int64_t d = shop_of[i] - shop_of[j];
                if (d < 0) d = -d;
                const double lo = d == 0 ? in_lo : in_hi * (double)d, hi = d == 0 ? in_hi : out_hi * (double)d;
                const double v = lo + (hi - lo) * mt.next_double();
                if (keep && j > i) { o[(size_t)i * M + j] = v; o[(size_t)j * M + i] = v; }
            }
    }
    *pos = mt.pos;
    return 0;
}
}
