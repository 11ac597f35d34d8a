/* A plain-C caller of include/culzss.h following the reference pipeline's call sequence
 * (cuda-lzss-cluster/culzss.c:85-180, deculzss.c:78-120): ring-slot buffers from initCPUmem /
 * initGPUmem, compression_kernel_wrapper -> onestream_finish_GPU -> aftercompression_wrapper per slot,
 * then decompression_kernel_wrapper in place.  The packed bytes are compared with the CPU oracle
 * (tests may link it), the decoded bytes with the input.  Built with gcc. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "culzss.h"

void orc_lzss_candidates(const uint8_t *in, int buf_length, uint8_t *out);
int orc_lzss_pack(const uint8_t *cand, int buf_length, uint8_t *packed, int *comp_length);

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

static void log_text(unsigned char *p, int n, uint32_t seed)
{   /* log-style lines: repetitive structure, varying fields */
    static const char *lvl[] = {"INFO", "WARN", "DEBUG", "ERROR"};
    int o = 0;
    while (o < n) {
        char line[160];
        const int k = snprintf(line, sizeof line, "2026-09-28T%02u:%02u:%02u.%03uZ host-%02u svc-%c[%u]: %s request id=%06x latency=%ums\n",
                               lcg(&seed) % 24, lcg(&seed) % 60, lcg(&seed) % 60, lcg(&seed) % 1000, lcg(&seed) % 16,
                               'a' + (char)(lcg(&seed) % 5), 1000 + lcg(&seed) % 9000, lvl[lcg(&seed) % 4], lcg(&seed) & 0xFFFFFF,
                               lcg(&seed) % 500);
        const int c = k < n - o ? k : n - o;
        memcpy(p + o, line, (size_t)c);
        o += c;
    }
}

int main(void)
{
    const int BUF = 1 << 20, NSLOT = 4;
    initGPU();                                                 /* culzss.c: queueInit */
    unsigned char *in_d = initGPUmem(BUF), *out_d = initGPUmem(BUF * 2);          /* culzss.c:85-86 */
    unsigned char *buf[4], *bufout[4], *orig[4];
    for (int s = 0; s < NSLOT; s++) {
        buf[s] = initCPUmem(BUF); bufout[s] = initCPUmem(BUF * 2); orig[s] = (unsigned char *)malloc((size_t)BUF);
        if (!buf[s] || !bufout[s] || !in_d || !out_d) { fprintf(stderr, "allocation failed\n"); return 2; }
        log_text(buf[s], BUF, 77u + (uint32_t)s);
        memcpy(orig[s], buf[s], (size_t)BUF);
    }
    for (int s = 0; s < NSLOT; s++)                            /* culzss.c:108 (gpu_consumer) */
        if (compression_kernel_wrapper(buf[s], BUF, bufout[s], 0, 0, 128, 0, s, in_d, out_d) != 1) return 1;
    uint8_t *cand = (uint8_t *)malloc((size_t)BUF * 2), *packed = (uint8_t *)malloc((size_t)BUF + 4096);
    int bad = 0;
    for (int s = 0; s < NSLOT; s++) {
        int comp = 0, want = 0;
        if (onestream_finish_GPU(s) != 1) return 1;            /* culzss.c:170 (cpu_consumer) */
        const int ok = aftercompression_wrapper(buf[s], BUF, bufout[s], &comp);       /* culzss.c:176 */
        orc_lzss_candidates(orig[s], BUF, cand);
        const int wok = orc_lzss_pack(cand, BUF, packed, &want);
        const int same_cand = memcmp(cand, bufout[s], (size_t)BUF * 2) == 0;
        const int same = ok == wok && (!ok || (comp == want && memcmp(packed, buf[s], (size_t)comp) == 0));
        printf("slot %d: ok=%d comp=%d oracle=%d candidates_equal=%d packed_equal=%d\n", s, ok, comp, want, same_cand, same);
        bad |= !same || !same_cand;
        if (ok) {                                              /* deculzss.c:98: in place */
            int dec = 0;
            if (decompression_kernel_wrapper(buf[s], comp, &dec, 0, 1, 1) != 1 || dec != BUF ||
                memcmp(buf[s], orig[s], (size_t)BUF) != 0) { printf("slot %d: round trip FAILED\n", s); bad = 1; }
        }
    }
    for (int s = 0; s < NSLOT; s++) { deleteCPUmem(buf[s]); deleteCPUmem(bufout[s]); free(orig[s]); }
    deleteGPUmem(in_d); deleteGPUmem(out_d); deleteGPUStreams();
    free(cand); free(packed);
    printf(bad ? "FAILED\n" : "ALL OK\n");
    return bad;
}
