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
            // the wire form of the rows (the multi-GPU exchange): packed into an exact-size block, unpacked and indexed from it, truncated
            // streams refused, the de-dup on the dense heads equal to the de-dup on the full rows
            if (cnt > 0) {
                const int64_t need = pm_packets_pack(rows.data(), cnt, nullptr, 0);
                if (need <= 0) return 4;
                std::vector<uint8_t> wire((size_t)need);
                if (pm_packets_pack(rows.data(), cnt, wire.data(), need) != need) return 5;
                std::vector<pm_packet> back((size_t)cnt);
                if (pm_packets_unpack(wire.data(), need, back.data(), cnt) != cnt) return 6;
                for (int64_t k = 0; k < cnt; ++k)
                    if (back[(size_t)k].streamaddress != rows[(size_t)k].streamaddress || back[(size_t)k].len != rows[(size_t)k].len ||
                        memcmp(back[(size_t)k].data, rows[(size_t)k].data, (size_t)rows[(size_t)k].len) != 0)
                        return 7;
                std::vector<pm_packet_head> heads((size_t)cnt);
                std::vector<int64_t> at((size_t)cnt);
                if (pm_packets_index(wire.data(), need, heads.data(), at.data(), cnt) != cnt) return 8;
                for (int64_t cut : {need - 1, need / 2, (int64_t)39, (int64_t)1}) {
                    if (cut <= 0 || cut >= need) continue;
                    std::vector<uint8_t> part(wire.begin(), wire.begin() + cut);          // exact size: reading past the cut is reading past the block
                    (void)pm_packets_unpack(part.data(), cut, back.data(), cnt);
                    (void)pm_packets_index(part.data(), cut, heads.data(), at.data(), cnt);
                }
                if (pm_packets_index(wire.data(), need, heads.data(), at.data(), cnt) != cnt) return 9;
                std::vector<int64_t> u1((size_t)cnt), u2((size_t)cnt);
                std::vector<int32_t> c1((size_t)cnt), c2((size_t)cnt);
                const int64_t counts[1] = {cnt};
                const int64_t k1 = pm_correlate(rows.data(), counts, 1, 1200.0, u1.data(), c1.data(), cnt);
                const int64_t k2 = pm_correlate_strided(heads.data(), (int64_t)sizeof(pm_packet_head), counts, 1, 1200.0, u2.data(), c2.data(), cnt);
                if (k1 < 0 || k1 != k2 || memcmp(u1.data(), u2.data(), (size_t)k1 * 8) != 0) return 10;
                mix(&k1, 8);
            }
            pm_codec_destroy(h);
        }
    }
    printf("%ld %016llx\n", total, (unsigned long long)digest);
    return 0;
}
