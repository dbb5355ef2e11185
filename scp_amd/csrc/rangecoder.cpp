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

struct scp_ac_dec {
    uint8_t *in = nullptr;
    size_t len = 0, ptr = 0;
    uint8_t cache = 0, cached_bits = 0;
    uint32_t low = 0, high = 0xFFFFFFFFu, value = 0;
    int32_t Lp = 0;
    inline void get() {
        if (cached_bits == 0) {
            if (ptr == len) { value <<= 1; return; }
            cache = in[ptr++];
            cached_bits = 8;
        }
        value = (value << 1) | ((cache >> (cached_bits - 1)) & 1u);
        --cached_bits;
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
    for (int i = 0; i < 32; ++i) p->get();
    *d = p;
    return SCP_OK;
}

extern "C" int scp_ac_dec_next(scp_ac_dec *d, const uint16_t *row) {
    if (!d || !row) return SCP_EINVAL;
    const int max_symbol = d->Lp - 2;
    const uint64_t span = (uint64_t)d->high - (uint64_t)d->low + 1;
    const uint16_t count = (uint16_t)((((uint64_t)d->value - (uint64_t)d->low + 1) * 0x10000ull - 1) / span);
    // largest s with cdf[s] <= count, searched over [0, max_symbol] like the reference's binsearch
    uint32_t left = 0, right = (uint32_t)max_symbol + 1;
    int sy = -1;
    while (left + 1 < right) {
        const uint32_t mid = (left + right) / 2;
        const uint16_t v = row[mid];
        if (v < count) left = mid; else if (v > count) right = mid; else { sy = (int)mid; break; }
    }
    if (sy < 0) sy = (int)left;
    const uint32_t c_low = row[sy], c_high = sy == max_symbol ? 0x10000u : (uint32_t)row[sy + 1];
    d->high = (d->low - 1) + (uint32_t)((span * (uint64_t)c_high) >> 16);
    d->low = d->low + (uint32_t)((span * (uint64_t)c_low) >> 16);
    for (;;) {
        if (d->low >= 0x80000000u || d->high < 0x80000000u) {
            d->low <<= 1; d->high = (d->high << 1) | 1u; d->get();
        } else if (d->low >= 0x40000000u && d->high < 0xC0000000u) {
            d->low = (d->low << 1) & 0x7FFFFFFFu; d->high = (d->high << 1) | 0x80000001u;
            d->value -= 0x40000000u; d->get();
        } else break;
    }
    return sy;
}

// decode n consecutive symbols, row i of the [n][Lp] table being the CDF of symbol i
extern "C" int scp_ac_dec_run(scp_ac_dec *d, const uint16_t *cdf, int64_t n, int16_t *out) {
    if (!d || !cdf || !out || n < 0) return SCP_EINVAL;
    for (int64_t i = 0; i < n; ++i) {
        const int s = scp_ac_dec_next(d, cdf + i * d->Lp);
        if (s < 0) return s;
        out[i] = (int16_t)s;
    }
    return SCP_OK;
}

extern "C" int scp_ac_dec_free(scp_ac_dec *d) {
    if (!d) return SCP_EINVAL;
    free(d->in);
    delete d;
    return SCP_OK;
}
