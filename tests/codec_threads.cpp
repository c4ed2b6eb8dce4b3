#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
#include "pymodem_amd.h"
int pm_set_error(int code, const char *fmt, ...) { (void)fmt; return code; }
int main()
{
    // four "host workers", each decoding recordings of eight chains through the shared pool (pm_pipe.hip's host stage without the GPU)
    std::vector<std::thread> ws;
    std::vector<long> totals(4, 0);
    for (int w = 0; w < 4; ++w)
        ws.emplace_back([w, &totals] {
            std::mt19937_64 rng(100 + w);
            for (int rec = 0; rec < 12; ++rec) {
                const int nch = 8;
                const int64_t n = 20000 + (int64_t)(rng() % 30000);
                std::vector<std::vector<uint8_t>> data(nch, std::vector<uint8_t>((size_t)n));
                std::vector<int64_t> addr((size_t)n);
                for (int64_t i = 0; i < n; ++i) addr[(size_t)i] = 40 * (i + 1);
                for (auto &d : data) for (auto &b : d) b = (uint8_t)(rng() >> 24);
                std::vector<pm_codec *> codecs(nch, nullptr);
                std::vector<pm_host_job> jobs(nch);
                for (int c = 0; c < nch; ++c) {
                    if (pm_codec_create(c % 2, 1, 0, 0, 2, c, &codecs[c])) std::abort();
                    memset(&jobs[c], 0, sizeof(pm_host_job));
                    jobs[c].codec = codecs[c]; jobs[c].h_data = data[c].data(); jobs[c].h_addr = addr.data(); jobs[c].n = n;
                    jobs[c].lfsr_poly = c % 3 ? 0x3 : 0x63003; jobs[c].lfsr_invert = c & 1;
                }
                if (pm_host_decode_batch(jobs.data(), nch, 8)) std::abort();
                std::vector<int64_t> counts(nch);
                int64_t total = 0;
                for (int c = 0; c < nch; ++c) total += (counts[c] = jobs[c].pending);
                std::vector<pm_packet> rows((size_t)(total > 0 ? total : 1));
                memset(rows.data(), 0, rows.size() * sizeof(pm_packet));
                if (pm_codec_fetch_batch(codecs.data(), counts.data(), nch, rows.data(), 8)) std::abort();
                std::vector<int64_t> uniq((size_t)(total > 0 ? total : 1));
                std::vector<int32_t> corr((size_t)(total > 0 ? total : 1));
                if (total && pm_correlate(rows.data(), counts.data(), nch, 1200.0, uniq.data(), corr.data(), total) < 0) std::abort();
                totals[w] += total;
                for (auto c : codecs) pm_codec_destroy(c);
            }
        });
    for (auto &t : ws) t.join();
    printf("%ld %ld %ld %ld\n", totals[0], totals[1], totals[2], totals[3]);
    return 0;
}
