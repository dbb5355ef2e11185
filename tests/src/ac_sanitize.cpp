#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "scp.h"
// random monotone CDF rows (incl. zero-width symbols never coded, peaky rows, non-monotone "other" rows are not valid for the coder), N symbols, round trip
int main() {
    srand(7);
    for (int trial = 0; trial < 200; ++trial) {
        const int Lp = 256, n = 1 + rand() % 5000;
        std::vector<uint16_t> cdf((size_t)n * Lp);
        std::vector<int16_t> sym(n), out(n);
        for (int i = 0; i < n; ++i) {
            // widths: 255 symbols, total 65536 -> cdf[255] = 0 (wraps), like the reference's int16 tables
            std::vector<uint32_t> w(255, 1);
            uint32_t left = 65536 - 255;
            const int mode = rand() % 3;
            while (left) { const int k = mode == 0 ? rand() % 255 : (mode == 1 ? (rand() % 4) * 60 : 254 - rand() % 3); const uint32_t a = 1 + rand() % (left < 4000 ? left : 4000); w[k] += a > left ? left : a; left -= a > left ? left : a; }
            uint32_t c = 0;
            for (int j = 0; j < 255; ++j) { cdf[(size_t)i * Lp + j] = (uint16_t)c; c += w[j]; }
            cdf[(size_t)i * Lp + 255] = (uint16_t)c;      // 65536 -> 0
            sym[i] = (int16_t)(rand() % 255);
        }
        std::vector<uint8_t> buf((size_t)n * 4 + 64);
        size_t len = 0;
        if (scp_ac_encode_cdf(cdf.data(), sym.data(), n, Lp, buf.data(), buf.size(), &len)) { printf("encode failed\n"); return 1; }
        std::vector<uint8_t> exact(buf.begin(), buf.begin() + len);          // exact-size copy: reads past the end show up under ASan
        scp_ac_dec *d = nullptr;
        if (scp_ac_dec_new(&d, exact.data(), exact.size(), Lp)) { printf("dec_new failed\n"); return 1; }
        const int half = n / 2;
        if (scp_ac_dec_run(d, cdf.data(), half, out.data())) { printf("run failed\n"); return 1; }
        for (int i = half; i < n; ++i) { const int s = scp_ac_dec_next(d, cdf.data() + (size_t)i * Lp); if (s < 0) { printf("next failed\n"); return 1; } out[i] = (int16_t)s; }
        scp_ac_dec_free(d);
        if (memcmp(out.data(), sym.data(), n * 2)) { printf("MISMATCH trial %d n %d\n", trial, n); return 1; }
    }
    printf("200 round trips ok\n");
    return 0;
}
