/* chain_demo.c -- one AFSK demod_chain driven from plain C through libpymodem_amd.so: modem + slicer on the GPU in one call
 * (pm_chain_run), then NRZI descrambling and AX.25 decoding on the host (pm_lfsr_unscramble, pm_codec_*).
 *
 *   gcc -std=c11 -I include examples/chain_demo.c -o chain_demo -L pymodem_amd -lpymodem_amd -Wl,-rpath,$PWD/pymodem_amd
 *   ./chain_demo taps.bin audio.s16
 *
 * taps.bin (written by the host that designed the filters, e.g. tests/test_c_example.py): little-endian
 *   int32 n_bpf, n_corr, n_lpf;  double samples_per_symbol, lock_rate;  then the doubles of
 *   input_bpf[n_bpf], mark_i[n_corr], mark_q[n_corr], space_i[n_corr], space_q[n_corr], output_lpf[n_lpf].
 * audio.s16: raw little-endian int16 samples.
 * Output: one line "bytes N", then one line per packet "packet <streamaddress> <len> <crc ok 0|1> <calculated crc>". */
#include "pymodem_amd.h"
#include <stdio.h>
#include <stdlib.h>

static void die(const char *what)
{
    char msg[512];
    pm_last_error(msg, sizeof msg);
    fprintf(stderr, "%s: %s\n", what, msg);
    exit(1);
}

static void *slurp(const char *path, size_t *bytes)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *p = malloc(n > 0 ? (size_t)n : 1);
    if (!p || fread(p, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    *bytes = (size_t)n;
    return p;
}

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s taps.bin audio.s16\n", argv[0]); return 2; }
    size_t tb, ab;
    char *t = slurp(argv[1], &tb);
    int16_t *audio = slurp(argv[2], &ab);
    const int64_t n = (int64_t)(ab / 2);
    const int32_t *hdr = (const int32_t *)t;
    const int n_bpf = hdr[0], n_corr = hdr[1], n_lpf = hdr[2];
    const double *d = (const double *)(t + 16);            /* three int32 + 4 bytes of padding */
    pm_chain_desc desc = {0};
    desc.modem = PM_MODEM_AFSK;
    desc.slicer.samples_per_symbol = d[0];
    desc.slicer.lock_rate = d[1];
    desc.slicer.bits_per_symbol = 1;
    desc.slicer.state_mask = 0x3;
    desc.slicer.demap[2] = desc.slicer.demap[3] = 1;       /* slicer.py:22-33: BinarySlicer */
    d += 2;
    desc.input_fir = d;  desc.n_input_fir = n_bpf;  d += n_bpf;
    desc.mark_i = d;  d += n_corr;
    desc.mark_q = d;  d += n_corr;
    desc.space_i = d; d += n_corr;
    desc.space_q = d; d += n_corr;
    desc.n_corr = n_corr;
    desc.output_fir = d; desc.n_output_fir = n_lpf;

    pm_ctx *ctx;
    pm_chain *chain;
    if (pm_ctx_create(0, &ctx)) die("pm_ctx_create");
    if (pm_chain_create(ctx, &desc, &chain)) die("pm_chain_create");
    const int64_t cap = n / 8 + 16;
    uint8_t *bytes = malloc((size_t)cap);
    int64_t *addrs = malloc((size_t)cap * sizeof(int64_t));
    int64_t count = 0;
    if (pm_chain_run(chain, audio, n, 0, bytes, addrs, cap, &count)) die("pm_chain_run");
    printf("bytes %lld\n", (long long)count);

    uint64_t shift_register = 0;
    uint8_t *plain = malloc((size_t)(count ? count : 1));
    if (pm_lfsr_unscramble(bytes, count, 0x3, 1, &shift_register, plain)) die("pm_lfsr_unscramble");   /* NRZI, inverted */
    pm_codec *codec;
    if (pm_codec_create(0, 0, 0, 0, 0, 0, &codec)) die("pm_codec_create");                              /* AX.25 */
    int64_t pending = 0, got = 0;
    if (pm_codec_decode(codec, plain, addrs, count, &pending)) die("pm_codec_decode");
    pm_packet *pk = malloc((size_t)(pending ? pending : 1) * sizeof(pm_packet));
    if (pm_codec_fetch(codec, pk, pending, &got)) die("pm_codec_fetch");
    for (int64_t k = 0; k < got; ++k)
        printf("packet %lld %d %d %d\n", (long long)pk[k].streamaddress, pk[k].len, pk[k].valid_crc, pk[k].calculated_crc);
    pm_codec_destroy(codec);
    pm_chain_destroy(chain);
    pm_ctx_destroy(ctx);
    free(pk); free(plain); free(bytes); free(addrs); free(audio); free(t);
    return 0;
}
