// 32-bit range coder with 16-bit integer CDFs (host, serial per stream).
//
// Drop-in for numpyAc/backend/numpyAc_backend.cpp: `encode` :245-323 (scp_ac_encode_cdf) and the stateful
// `class decode` :134-217 (scp_ac_dec_*).  scp_ac_encode_lohi consumes the (c_low, c_high) pairs the device
// CDF kernel emits, so the 512 B/symbol CDF table never crosses PCIe.  Bit-exact with the reference
// (tests/test_abi.py and tests/test_gpu_e2e.py against tests/golden/ac_streams.npz).
//
// Output is produced through a 64-bit accumulator and written a byte at a time; pending (underflow) bits
// are flushed in whole-word bursts instead of the reference's bit-by-bit std::string appends.
#include <stdlib.h>
#include <string.h>
#include <new>
#include <type_traits>
#include "../../include/scp.h"

namespace {

struct BitSink {
    uint8_t *out;
    size_t cap, len = 0;
    uint64_t acc = 0;  // bits collected MSB-first in the low `nbits` bits
    int nbits = 0;
    bool overflow = false;
    BitSink(uint8_t *o, size_t c) : out(o), cap(c) {}
    inline void drain() {
        while (nbits >= 8) {
            const uint8_t b = (uint8_t)(acc >> (nbits - 8));
            if (len < cap) out[len] = b; else overflow = true;
            ++len;
            nbits -= 8;
        }
    }
    inline void put(uint32_t bit) { acc = (acc << 1) | bit; if (++nbits >= 32) drain(); }
    inline void put_run(uint32_t bit, uint64_t count) {  // `count` copies of `bit`
        while (count) {
            drain();  // nbits < 8 from here, so 32 more bits always fit
            const int k = count > 32 ? 32 : (int)count;
            acc = (acc << k) | (bit ? ((1ull << k) - 1ull) : 0ull);
            nbits += k;
            count -= k;
            drain();
        }
    }
    inline void put_with_pending(uint32_t bit, uint64_t &pending) {
        put(bit);
        if (pending) { put_run(bit ^ 1u, pending); pending = 0; }
    }
    inline void put_bits(uint32_t v, int k) {   // the low k (<= 31) bits of v, MSB first
        drain();                                // nbits < 8 from here
        acc = (acc << k) | (uint64_t)v;
        nbits += k;
        drain();
    }
    inline void finish() {
        drain();
        if (nbits) { acc <<= (8 - nbits); nbits = 8; drain(); }  // zero-pad the last byte
    }
};

struct Coder {
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint64_t pending = 0;
    inline void step(BitSink &s, uint32_t c_low, uint32_t c_high) {
        const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        high = (low - 1) + (uint32_t)((span * (uint64_t)c_high) >> 16);
        low = low + (uint32_t)((span * (uint64_t)c_low) >> 16);
        // Renormalisation, all shifts of a kind at once (the reference shifts one bit per loop iteration; the three cases cannot
        // interleave: once the top bits differ only underflow shifts remain).
        // (1) k leading bits shared by low and high leave as output bits: the first one followed by the pending (underflow)
        //     bits, the other k - 1 verbatim.  low < high always (c_high > c_low, span > 2^30), so k <= 31.
        const uint32_t diff = low ^ high;
        if (!(diff & 0x80000000u)) {
            const int k = __builtin_clz(diff);
            const uint32_t top = low >> (32 - k);             // the k shared bits
            s.put_with_pending(top >> (k - 1), pending);
            if (k > 1) s.put_bits(top & ((1u << (k - 1)) - 1u), k - 1);
            low <<= k;
            high = (high << k) | ((1u << k) - 1u);
        }
        // (2) underflow: low = 01..., high = 10...: one shift per position where low continues with 1s and high with 0s
        if (low >= 0x40000000u && high < 0xC0000000u) {
            const uint32_t l1 = ~(low << 1), h1 = high << 1;  // leading 1s of low / leading 0s of high after the top bit
            const int a = l1 ? __builtin_clz(l1) : 32, b = h1 ? __builtin_clz(h1) : 32;
            int u = a < b ? a : b;                            // >= 1 here
            if (u > 31) u = 31;
            pending += (uint64_t)u;
            low = (low << u) & 0x7FFFFFFFu;
            high = (high << u) | 0x80000000u | ((1u << u) - 1u);
        }
    }
    inline void flush(BitSink &s) {
        pending += 1;
        s.put_with_pending(low < 0x40000000u ? 0u : 1u, pending);
        s.finish();
    }
};

}  // namespace

extern "C" int scp_ac_encode_cdf(const uint16_t *cdf, const int16_t *sym, int64_t n, int32_t Lp, uint8_t *out, size_t cap,
                                 size_t *out_len) {
    if (!cdf || !sym || !out || !out_len || n < 0 || Lp < 2) return SCP_EINVAL;
    BitSink s(out, cap);
    Coder c;
    const int max_symbol = Lp - 2;
    for (int64_t i = 0; i < n; ++i) {
        const int sy = sym[i];
        if (sy < 0 || sy > max_symbol) return SCP_EINVAL;
        const uint16_t *row = cdf + i * Lp;
        c.step(s, row[sy], sy == max_symbol ? 0x10000u : (uint32_t)row[sy + 1]);
    }
    c.flush(s);
    *out_len = s.len;
    return s.overflow ? SCP_ESMALL : SCP_OK;
}

extern "C" int scp_ac_encode_lohi(const uint32_t *lohi, int64_t n, uint8_t *out, size_t cap, size_t *out_len) {
    if (!lohi || !out || !out_len || n < 0) return SCP_EINVAL;
    BitSink s(out, cap);
    Coder c;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t v = lohi[i];
        const uint32_t hi = v >> 16;
        c.step(s, v & 0xFFFFu, hi ? hi : 0x10000u);
    }
    c.flush(s);
    *out_len = s.len;
    return s.overflow ? SCP_ESMALL : SCP_OK;
}

// Decoder state.  The reference (numpyAc_backend.cpp:134-217) shifts `low`, `high` and `value` one bit per loop iteration and fetches the
// next stream bit through a byte cache; here the bits wait in a 64-bit window and every renormalisation is two bursts, the mirror image of
// Coder::step above: (1) the k leading bits low and high share leave together, (2) the u underflow positions (low = 01.., high = 10..) leave
// together - u single steps of value = ((value - 2^30) << 1) | bit amount to (value << u) with the top bit flipped, plus u new bits.  The
// cases cannot interleave (after (2) the top bits differ and no underflow position is left), so a symbol costs at most two bursts.
// Bits behind the end of the stream read as zero, like the reference's get().
struct scp_ac_dec {
    uint8_t *in = nullptr;
    size_t len = 0, ptr = 0;
    uint64_t win = 0;       // the next `nwin` stream bits, left-aligned (bit 63 first)
    int nwin = 0;
    uint32_t low = 0, high = 0xFFFFFFFFu, value = 0;
    int32_t Lp = 0;
    inline void refill() {  // keep at least 32 bits in the window (zeros behind the end of the stream)
        while (nwin <= 56) {
            const uint64_t b = ptr < len ? in[ptr] : 0u;
            ++ptr;
            win |= b << (56 - nwin);
            nwin += 8;
        }
    }
    inline uint32_t take(int k) {   // the next k (0 .. 31) bits, MSB first
        if (k <= 0) return 0;       // (tables with zero-width symbols can reach low > high and ask for no bits: win >> 64 is undefined)
        if (nwin < k) refill();
        const uint32_t v = (uint32_t)(win >> (64 - k));
        win <<= k;
        nwin -= k;
        return v;
    }
};

extern "C" int scp_ac_dec_new(scp_ac_dec **d, const uint8_t *stream, size_t len, int32_t Lp) {
    if (!d || (!stream && len) || Lp < 2) return SCP_EINVAL;
    scp_ac_dec *p = new (std::nothrow) scp_ac_dec();
    if (!p) return SCP_ENOMEM;
    p->in = (uint8_t *)malloc(len ? len : 1);
    if (!p->in) { delete p; return SCP_ENOMEM; }
    if (len) memcpy(p->in, stream, len);
    p->len = len;
    p->Lp = Lp;
    p->refill();
    p->value = (p->take(16) << 16) | p->take(16);
    *d = p;
    return SCP_OK;
}

// largest s with cdf[s] <= count, searched over [0, max_symbol] exactly like the reference's binsearch (numpyAc_backend.cpp:100-131)
static inline int ac_search_ref(const uint16_t *row, uint16_t count, int max_symbol) {
    uint32_t left = 0, right = (uint32_t)max_symbol + 1;
    while (left + 1 < right) {
        const uint32_t mid = (left + right) / 2;
        const uint16_t v = row[mid];
        if (v < count) left = mid; else if (v > count) right = mid; else return (int)mid;
    }
    return (int)left;
}

#if defined(__x86_64__)
#include <immintrin.h>
// 256-entry rows (255 symbols): on a non-decreasing row the answer is (number of entries <= count among the first 255) - 1: sixteen
// unsigned 16-bit vector compares and popcounts, no data-dependent branch (the binary search mispredicts every other step).  The result is
// CHECKED (row[s] <= count < row[s + 1]); a row that is not monotone there falls back to the reference's search, so any table decodes as before.
__attribute__((target("avx2,popcnt"))) static inline int ac_search_256_avx2(const uint16_t *row, uint16_t count) {
    const __m256i c = _mm256_set1_epi16((short)count);
    unsigned le = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(row + 16 * i));
        const __m256i m = _mm256_cmpeq_epi16(_mm256_min_epu16(v, c), v);          // v <= count (unsigned)
        unsigned bits = (unsigned)_mm256_movemask_epi8(m);                        // two bits per entry
        if (i == 15) bits &= 0x3FFFFFFFu;                                         // entry 255 is not a symbol
        le += (unsigned)_mm_popcnt_u32(bits);
    }
    const int s = (int)(le >> 1) - 1;
    if (s < 0 || row[s] > count || (s < 254 && row[s + 1] <= count)) return -1;
    if (s > 0 && row[s - 1] == row[s]) return -1;    // zero-width neighbours: the reference's search decides which of the equal entries it hits
    return s;
}
static const bool g_ac_avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt");
#endif

template <bool FAST256>
static inline int ac_dec_step(scp_ac_dec *d, const uint16_t *row, int max_symbol) {
    uint32_t low = d->low, high = d->high, value = d->value;
    const uint64_t span = (uint64_t)high - (uint64_t)low + 1;
    const uint16_t count = (uint16_t)((((uint64_t)value - (uint64_t)low + 1) * 0x10000ull - 1) / span);
    int sy = -1;
#if defined(__x86_64__)
    if (FAST256) sy = ac_search_256_avx2(row, count);
#endif
    if (sy < 0) sy = ac_search_ref(row, count, max_symbol);
    const uint32_t c_low = row[sy], c_high = sy == max_symbol ? 0x10000u : (uint32_t)row[sy + 1];
    high = (low - 1) + (uint32_t)((span * (uint64_t)c_high) >> 16);
    low = low + (uint32_t)((span * (uint64_t)c_low) >> 16);
    const uint32_t diff = low ^ high;
    if (!(diff & 0x80000000u)) {                           // (1) k shared leading bits (k <= 31: low < high)
        const int k = diff ? __builtin_clz(diff) : 31;   // (diff == 0 only on tables with zero-width symbols)
        low <<= k;
        high = (high << k) | ((1u << k) - 1u);
        value = (value << k) | d->take(k);
    }
    if (low >= 0x40000000u && high < 0xC0000000u) {        // (2) u underflow positions
        const uint32_t l1 = ~(low << 1), h1 = high << 1;
        const int a = l1 ? __builtin_clz(l1) : 32, b = h1 ? __builtin_clz(h1) : 32;
        int u = a < b ? a : b;
        if (u > 31) u = 31;
        low = (low << u) & 0x7FFFFFFFu;
        high = (high << u) | 0x80000000u | ((1u << u) - 1u);
        value = ((value << u) ^ 0x80000000u) | d->take(u);
    }
    d->low = low; d->high = high; d->value = value;
    return sy;
}

extern "C" int scp_ac_dec_next(scp_ac_dec *d, const uint16_t *row) {
    if (!d || !row) return SCP_EINVAL;
    return ac_dec_step<false>(d, row, d->Lp - 2);
}

// decode n consecutive symbols, row i of the [n][Lp] table being the CDF of symbol i
extern "C" int scp_ac_dec_run(scp_ac_dec *d, const uint16_t *cdf, int64_t n, int16_t *out) {
    if (!d || !cdf || !out || n < 0) return SCP_EINVAL;
    const int Lp = d->Lp, max_symbol = Lp - 2;
    // the table arrives by DMA (pinned memory, not in any cache) and the search touches 4 - 5 scattered lines of a 512-byte row: the rows a
    // few symbols ahead are prefetched whole (without it a symbol costs two to three exposed memory latencies)
    const int64_t ahead = 6;
    const int lines = (Lp * 2 + 63) / 64;
    bool fast = false;
#if defined(__x86_64__)
    fast = g_ac_avx2 && Lp == 256;
#endif
    auto run = [&](auto tag) {
        constexpr bool F = decltype(tag)::value;
        for (int64_t i = 0; i < n; ++i) {
            if (i + ahead < n) {
                const char *p = (const char *)(cdf + (i + ahead) * Lp);
                for (int l = 0; l < lines; ++l) __builtin_prefetch(p + 64 * l, 0, 0);
            }
            out[i] = (int16_t)ac_dec_step<F>(d, cdf + i * Lp, max_symbol);
        }
    };
    if (fast) run(std::true_type()); else run(std::false_type());
    return SCP_OK;
}

extern "C" int scp_ac_dec_free(scp_ac_dec *d) {
    if (!d) return SCP_EINVAL;
    free(d->in);
    delete d;
    return SCP_OK;
}
