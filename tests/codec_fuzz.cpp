// ASan/UBSan harness for the host codecs: random streams, random call sizes, digest of what comes out
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "pymodem_amd.h"
int pm_set_error(int code, const char *fmt, ...) { (void)fmt; return code; }
int main()
{
    std::mt19937_64 rng(77);
    uint64_t digest = 1469598103934665603ull;
    long total = 0;
    auto mix = [&](const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; ++i) { digest ^= b[i]; digest *= 1099511628211ull; } };
    for (int trial = 0; trial < 400; ++trial) {
        const int64_t n = 30 + (int64_t)(rng() % 120000);
        const double ps[] = {0.5, 0.55, 0.7, 0.85, 0.45, 0.3, 0.93, 0.1};
        const double p = ps[trial % 8];
        std::vector<uint8_t> data((size_t)n);
        for (auto &b : data) { unsigned v = 0; for (int i = 0; i < 8; ++i) v = v << 1 | ((rng() >> 11) * (1.0 / 9007199254740992.0) < p); b = (uint8_t)v; }
        if (trial % 8 >= 4)
            for (int64_t k = 0; k < n / 300 + 1; ++k) data[(size_t)(rng() % (uint64_t)n)] = 0x7E;
        std::vector<int64_t> addr((size_t)n);
        int64_t a = 0;
        for (auto &x : addr) x = (a += 1 + (int64_t)(rng() % 50));
        for (int kind = 0; kind < 2; ++kind) {
            pm_codec *h = nullptr;
            if (pm_codec_create(kind, 1, 0, 0, 2, 0, &h)) return 1;
            const int64_t tops[] = {n + 1, 5000, 300, 40, 9};
            const int64_t top = tops[rng() % 5];
            int64_t pend = 0;
            // exact-size heap copies per call: a read past the end of a piece is a read past an allocation
            for (int64_t at = 0; at < n;) {
                int64_t step = 1 + (int64_t)(rng() % (uint64_t)top);
                if (step > n - at) step = n - at;
                std::vector<uint8_t> d(data.begin() + at, data.begin() + at + step);
                std::vector<int64_t> ad(addr.begin() + at, addr.begin() + at + step);
                if (pm_codec_decode(h, d.data(), ad.data(), step, &pend)) return 2;
                at += step;
            }
            std::vector<pm_packet> rows((size_t)(pend > 0 ? pend : 1));
            int64_t cnt = 0;
            if (pm_codec_fetch(h, rows.data(), pend, &cnt)) return 3;
            for (int64_t k = 0; k < cnt; ++k) { mix(rows[(size_t)k].data, (size_t)rows[(size_t)k].len); mix(&rows[(size_t)k].streamaddress, 8); mix(&rows[(size_t)k].calculated_crc, 4); }
            total += cnt;
            pm_codec_destroy(h);
        }
    }
    printf("%ld %016llx\n", total, (unsigned long long)digest);
    return 0;
}
